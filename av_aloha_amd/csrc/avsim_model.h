// avsim_model.h -- host-side parsing of a compiled .avm model blob (av_aloha_amd/compiler/compile.py)
// and the device-resident model image the kernels read.  Replaces the MuJoCo model the reference
// builds at env.py:53-56.
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace avs {

struct BlobEntry {
    char name[32];
    uint32_t dtype, ndim, dims[4];
    uint64_t offset, nbytes;
} __attribute__((packed));

class Blob {
  public:
    Blob(const void* p, size_t n) : data_((const char*)p, (const char*)p + n) {
        if (n < 16 || std::memcmp(data_.data(), "AVSIMMDL", 8) != 0) throw std::runtime_error("bad model blob magic");
        std::memcpy(&count_, data_.data() + 12, 4);
        if (16 + (size_t)count_ * sizeof(BlobEntry) > n) throw std::runtime_error("truncated model blob");
    }
    const BlobEntry& find(const char* name) const {
        const BlobEntry* e = (const BlobEntry*)(data_.data() + 16);
        for (uint32_t i = 0; i < count_; i++)
            if (std::strncmp(e[i].name, name, 32) == 0) {
                if (e[i].offset + e[i].nbytes > data_.size()) throw std::runtime_error("blob entry out of range");
                return e[i];
            }
        throw std::runtime_error(std::string("model blob lacks array ") + name);
    }
    std::vector<double> f(const char* name) const {
        const BlobEntry& e = find(name);
        if (e.dtype != 0) throw std::runtime_error(std::string("not f64: ") + name);
        std::vector<double> v(e.nbytes / 8);
        std::memcpy(v.data(), data_.data() + e.offset, e.nbytes);
        return v;
    }
    std::vector<int> i(const char* name) const {
        const BlobEntry& e = find(name);
        if (e.dtype != 1) throw std::runtime_error(std::string("not i32: ") + name);
        std::vector<int> v(e.nbytes / 4);
        std::memcpy(v.data(), data_.data() + e.offset, e.nbytes);
        return v;
    }
    int scalar(const char* name) const { return i(name).at(0); }

  private:
    std::vector<char> data_;
    uint32_t count_ = 0;
};

// IK constants of one arm (kinematics.py:7-15, 28-33 evaluated at the zero pose by the model compiler)
struct IkArm {
    int n;
    double w[7][3], v[7][3];  // screw axes: v = -w x p0
    double site0[12];         // rows 0..2 of the 4x4 home pose of the eef site
    double lo[7], hi[7];      // joint ranges
    int qadr[7];              // qpos address of each joint
};

struct IkParams {
    IkArm arm[3];
    // DiffIK (sim_env.py:125-138); for the manipulators the same gains with q0 = home (build choice)
    double k_pos, k_ori, damping, max_angvel, dt;
    double k_null[3][7], q0[3][7];
    int diff_iters;
    // GradIK (sim_env.py:89-122)
    double g_step, g_min_delta, g_pw, g_rw, g_pthr, g_rthr, g_maxp, g_maxr, g_joint_p;
    double g_jcw[6], g_jdw[6];
    int grad_iters;
    double grip_lo, grip_hi;
};

}  // namespace avs
