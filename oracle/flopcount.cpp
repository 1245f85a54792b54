// flopcount.cpp -- FLOP-census build of the oracle: the same lcr_oracle.c, compiled as C++ with an instrumented scalar.
// Used only by tools/count_flops.py (measurement infrastructure); nothing in the product loads it.
#define ORC_COUNT 1
#include "orc_counted.hpp"
orc_counts g_orc_counts = {0, 0, 0, 0, 0, 0, 0};
// libm calls made on plain doubles (the source casts `(real)sqrt((double)x)`) are counted through these wrappers
#include <math.h>
#include <stdlib.h>
#include <string.h>
static inline double cnt_sqrt(double x) { g_orc_counts.sqrt_++; return std::sqrt(x); }
static inline double cnt_sin(double x) { g_orc_counts.trans++; return std::sin(x); }
static inline double cnt_cos(double x) { g_orc_counts.trans++; return std::cos(x); }
static inline double cnt_tan(double x) { g_orc_counts.trans++; return std::tan(x); }
static inline double cnt_exp(double x) { g_orc_counts.trans++; return std::exp(x); }
static inline double cnt_pow(double x, double y) { g_orc_counts.trans++; return std::pow(x, y); }
static inline double cnt_fabs(double x) { g_orc_counts.abs_minmax++; return std::fabs(x); }
static inline double cnt_floor(double x) { g_orc_counts.abs_minmax++; return std::floor(x); }
#define sqrt cnt_sqrt
#define sin cnt_sin
#define cos cnt_cos
#define tan cnt_tan
#define exp cnt_exp
#define pow cnt_pow
#define fabs cnt_fabs
#define floor cnt_floor
#include "lcr_oracle.c"
extern "C" void orc_count_reset(void) { g_orc_counts = orc_counts{0, 0, 0, 0, 0, 0, 0}; }
extern "C" void orc_count_get(uint64_t out[7]) {
    out[0] = g_orc_counts.add; out[1] = g_orc_counts.mul; out[2] = g_orc_counts.div; out[3] = g_orc_counts.sqrt_;
    out[4] = g_orc_counts.trans; out[5] = g_orc_counts.cmp; out[6] = g_orc_counts.abs_minmax;
}
