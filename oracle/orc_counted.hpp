// orc_counted.hpp -- instrumented scalar for the FLOP CENSUS build of the CPU oracle (test/measurement infrastructure).
// liblcr_oracle_count.so = lcr_oracle.c compiled as C++ with orc_real = orc_counted: every arithmetic operation of the
// restated algorithm increments a counter, which gives the "algorithmic flops per env-step" figure of SURVEY.md 8(d).
#pragma once
#include <cmath>
#include <cstdint>
#include <type_traits>

struct orc_counts { uint64_t add, mul, div, sqrt_, trans, cmp, abs_minmax; };
extern orc_counts g_orc_counts;

struct orc_counted {
    double v;
    orc_counted() = default;
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
    orc_counted(T x) : v((double)x) {}
    template <class T, class = typename std::enable_if<std::is_arithmetic<T>::value>::type>
    explicit operator T() const { return (T)v; }
    orc_counted operator-() const { return orc_counted(-v); }
    orc_counted &operator+=(orc_counted o) { g_orc_counts.add++; v += o.v; return *this; }
    orc_counted &operator-=(orc_counted o) { g_orc_counts.add++; v -= o.v; return *this; }
    orc_counted &operator*=(orc_counted o) { g_orc_counts.mul++; v *= o.v; return *this; }
    orc_counted &operator/=(orc_counted o) { g_orc_counts.div++; v /= o.v; return *this; }
};
#define ORC_ARITH(T) typename std::enable_if<std::is_arithmetic<T>::value, int>::type = 0
#define ORC_BINOP(op, ctr)                                                                                              \
    inline orc_counted operator op(orc_counted a, orc_counted b) { g_orc_counts.ctr++; return orc_counted(a.v op b.v); } \
    template <class T, ORC_ARITH(T)> inline orc_counted operator op(orc_counted a, T b) { g_orc_counts.ctr++; return orc_counted(a.v op (double)b); } \
    template <class T, ORC_ARITH(T)> inline orc_counted operator op(T a, orc_counted b) { g_orc_counts.ctr++; return orc_counted((double)a op b.v); }
ORC_BINOP(+, add)
ORC_BINOP(-, add)
ORC_BINOP(*, mul)
ORC_BINOP(/, div)
#define ORC_CMPOP(op)                                                                                   \
    inline bool operator op(orc_counted a, orc_counted b) { g_orc_counts.cmp++; return a.v op b.v; }    \
    template <class T, ORC_ARITH(T)> inline bool operator op(orc_counted a, T b) { g_orc_counts.cmp++; return a.v op (double)b; } \
    template <class T, ORC_ARITH(T)> inline bool operator op(T a, orc_counted b) { g_orc_counts.cmp++; return (double)a op b.v; }
ORC_CMPOP(<)
ORC_CMPOP(>)
ORC_CMPOP(<=)
ORC_CMPOP(>=)
ORC_CMPOP(==)
ORC_CMPOP(!=)
inline orc_counted sqrt(orc_counted a) { g_orc_counts.sqrt_++; return orc_counted(std::sqrt(a.v)); }
inline orc_counted fabs(orc_counted a) { g_orc_counts.abs_minmax++; return orc_counted(std::fabs(a.v)); }
inline orc_counted floor(orc_counted a) { g_orc_counts.abs_minmax++; return orc_counted(std::floor(a.v)); }
inline orc_counted sin(orc_counted a) { g_orc_counts.trans++; return orc_counted(std::sin(a.v)); }
inline orc_counted cos(orc_counted a) { g_orc_counts.trans++; return orc_counted(std::cos(a.v)); }
inline orc_counted tan(orc_counted a) { g_orc_counts.trans++; return orc_counted(std::tan(a.v)); }
inline orc_counted exp(orc_counted a) { g_orc_counts.trans++; return orc_counted(std::exp(a.v)); }
inline orc_counted pow(orc_counted a, orc_counted b) { g_orc_counts.trans++; return orc_counted(std::pow(a.v, b.v)); }
template <class T, ORC_ARITH(T)> inline orc_counted pow(orc_counted a, T b) { g_orc_counts.trans++; return orc_counted(std::pow(a.v, (double)b)); }
template <class T, ORC_ARITH(T)> inline orc_counted pow(T a, orc_counted b) { g_orc_counts.trans++; return orc_counted(std::pow((double)a, b.v)); }
