// cdouble.h -- a double that counts its arithmetic.  The flop-counting build of the oracle (tools/count_flops.py) compiles every
// oracle/orc_*.c as C++ with `#define double cdouble`: +, -, *, / and sqrt count one flop each, the libm transcendentals eight,
// comparisons / abs / min / max / negation none; the counts go to orc_flops[orc_phase] (phases set by ORC_PHASE in orc_dyn.c).
// Test / measurement infrastructure only.
#pragma once
#include <cmath>
#include <cstring>
#include <cstdlib>
extern "C" { extern long long orc_flops[16]; extern int orc_phase; }
struct cdouble {
    double v;
    cdouble() = default;
    cdouble(double x) : v(x) {}
    cdouble(int x) : v(x) {}
    cdouble(long x) : v(x) {}
    cdouble(float x) : v(x) {}
    cdouble(unsigned x) : v(x) {}
    explicit operator double() const { return v; }
    explicit operator int() const { return (int)v; }
    explicit operator float() const { return (float)v; }
    explicit operator bool() const { return v != 0; }
    cdouble& operator+=(cdouble o) { orc_flops[orc_phase]++; v += o.v; return *this; }
    cdouble& operator-=(cdouble o) { orc_flops[orc_phase]++; v -= o.v; return *this; }
    cdouble& operator*=(cdouble o) { orc_flops[orc_phase]++; v *= o.v; return *this; }
    cdouble& operator/=(cdouble o) { orc_flops[orc_phase]++; v /= o.v; return *this; }
    cdouble operator-() const { return cdouble(-v); }
};
#define CD_BIN(op) \
  inline cdouble operator op(cdouble a, cdouble b) { orc_flops[orc_phase]++; return cdouble(a.v op b.v); } \
  inline cdouble operator op(cdouble a, double b) { orc_flops[orc_phase]++; return cdouble(a.v op b); } \
  inline cdouble operator op(double a, cdouble b) { orc_flops[orc_phase]++; return cdouble(a op b.v); } \
  inline cdouble operator op(cdouble a, int b) { orc_flops[orc_phase]++; return cdouble(a.v op b); } \
  inline cdouble operator op(int a, cdouble b) { orc_flops[orc_phase]++; return cdouble(a op b.v); }
CD_BIN(+) CD_BIN(-) CD_BIN(*) CD_BIN(/)
#define CD_CMP(op) \
  inline bool operator op(cdouble a, cdouble b) { return a.v op b.v; } \
  inline bool operator op(cdouble a, double b) { return a.v op b; } \
  inline bool operator op(double a, cdouble b) { return a op b.v; } \
  inline bool operator op(cdouble a, int b) { return a.v op b; } \
  inline bool operator op(int a, cdouble b) { return a op b.v; }
CD_CMP(<) CD_CMP(>) CD_CMP(<=) CD_CMP(>=) CD_CMP(==) CD_CMP(!=)
inline cdouble sqrt(cdouble a) { orc_flops[orc_phase]++; return cdouble(std::sqrt(a.v)); }
inline cdouble fabs(cdouble a) { return cdouble(std::fabs(a.v)); }
inline cdouble sin(cdouble a) { orc_flops[orc_phase] += 8; return cdouble(std::sin(a.v)); }
inline cdouble cos(cdouble a) { orc_flops[orc_phase] += 8; return cdouble(std::cos(a.v)); }
inline cdouble tan(cdouble a) { orc_flops[orc_phase] += 8; return cdouble(std::tan(a.v)); }
inline cdouble acos(cdouble a) { orc_flops[orc_phase] += 8; return cdouble(std::acos(a.v)); }
inline cdouble asin(cdouble a) { orc_flops[orc_phase] += 8; return cdouble(std::asin(a.v)); }
inline cdouble atan2(cdouble a, cdouble b) { orc_flops[orc_phase] += 8; return cdouble(std::atan2(a.v, b.v)); }
inline cdouble pow(cdouble a, cdouble b) { orc_flops[orc_phase] += 8; return cdouble(std::pow(a.v, b.v)); }
inline cdouble exp(cdouble a) { orc_flops[orc_phase] += 8; return cdouble(std::exp(a.v)); }
inline cdouble fmax(cdouble a, cdouble b) { return cdouble(std::fmax(a.v, b.v)); }
inline cdouble fmin(cdouble a, cdouble b) { return cdouble(std::fmin(a.v, b.v)); }
inline cdouble floor(cdouble a) { return cdouble(std::floor(a.v)); }
inline bool isfinite(cdouble a) { return std::isfinite(a.v); }
inline bool isnan(cdouble a) { return std::isnan(a.v); }
inline cdouble fmax(double a, cdouble b) { return cdouble(std::fmax(a, b.v)); }
inline cdouble fmax(cdouble a, double b) { return cdouble(std::fmax(a.v, b)); }
inline cdouble fmin(double a, cdouble b) { return cdouble(std::fmin(a, b.v)); }
inline cdouble fmin(cdouble a, double b) { return cdouble(std::fmin(a.v, b)); }
inline cdouble pow(cdouble a, double b) { orc_flops[orc_phase] += 8; return cdouble(std::pow(a.v, b)); }
inline cdouble pow(cdouble a, int b) { orc_flops[orc_phase] += 8; return cdouble(std::pow(a.v, b)); }
inline cdouble atan2(cdouble a, double b) { orc_flops[orc_phase] += 8; return cdouble(std::atan2(a.v, b)); }
inline cdouble atan2(double a, cdouble b) { orc_flops[orc_phase] += 8; return cdouble(std::atan2(a, b.v)); }
