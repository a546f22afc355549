// avsim_phys_f64.hip -- the double-precision instantiation of the physics kernel (AVSIM_F64_PHYSICS, the parity mode).
// Its own translation unit so that it can be compiled with -ffp-contract=off (av_aloha_amd/build.py): no a*b+c is fused, every
// expression rounds as in the oracle (oracle/Makefile compiles with -ffp-contract=off), and the discrete decisions of the narrow
// phase -- which hull vertex supports a direction, which clipped vertex is kept, where the portal walk of MPR turns -- fall the
// same way on both sides.  The f32 product kernel stays in avsim_api.hip with contraction on.
#define AVSIM_TU_F64 1
#include "avsim_phys.hip.h"

namespace avs {

int phys_launch_f64(PhysHost& ph, hipStream_t st, int nsub, const float* action, void* qpos, void* qvel, void* ctrl, void* warm, int* latch,
                    double* agent, int32_t* reward, uint8_t* success, std::string& err) {
    // double precision doubles the LDS record and the registers (492 unified VGPRs: one wave per SIMD): as many envs (wavefronts) per
    // workgroup as fit next to ONE copy of the tables in 160 KiB, at most four -- three for the 35 KB records, where one env per
    // workgroup with its own table copy fitted two per CU
    return ph.launch_t<double, 64, 4>(st, ph.md, nsub, action, qpos, qvel, ctrl, warm, latch, agent, reward, success, err);
}

}  // namespace avs
