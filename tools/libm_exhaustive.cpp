// Every float through every restated libm function of kernels/refmath.h against the host libm (two-argument functions along
// a few slices).  g++ -std=c++17 -O2 -ffp-contract=off -mfma -pthread -o /tmp/exh tools/libm_exhaustive.cpp -lm; ~2 min on 8 cores.
#include "../goblin_amd/csrc/kernels/refmath.h"
#include <cstdio>
#include <thread>
#include <vector>
#include <atomic>
typedef float (*fn1)(float);
static float g_exp(float x){return gbl_expf(x);} static float g_log(float x){return gbl_logf(x);} static float g_log2(float x){return gbl_log2f(x);}
static float g_atan(float x){return gbl_atanf(x);} static float g_tan(float x){return gbl_tanf(x);} static float g_acos(float x){return gbl_acosf(x);}
static float g_sin(float x){return gbl_sinf(x);} static float g_cos(float x){return gbl_cosf(x);}
static float g_pow15(float x){return gbl_powf(x,1.5f);} static float l_pow15(float x){return powf(x,1.5f);}
static float g_pow25(float x){return gbl_powf(x,25.0f);} static float l_pow25(float x){return powf(x,25.0f);}
static float g_powi(float x){return gbl_powf(x,1.0f/3.5f);} static float l_powi(float x){return powf(x,1.0f/3.5f);}
static float g_atan2a(float x){return gbl_atan2f(x,0.75f);} static float l_atan2a(float x){return atan2f(x,0.75f);}
static float g_atan2b(float x){return gbl_atan2f(-1.25f,x);} static float l_atan2b(float x){return atan2f(-1.25f,x);}
int main() {
    struct { const char* name; fn1 a, b; bool lim; } F[] = {{"exp", expf, g_exp, false}, {"log", logf, g_log, false}, {"log2", log2f, g_log2, false}, {"atan", atanf, g_atan, false},
        {"tan", tanf, g_tan, false}, {"acos", acosf, g_acos, false}, {"sin<120", sinf, g_sin, true}, {"cos<120", cosf, g_cos, true},
        {"pow(x,1.5)", l_pow15, g_pow15, false}, {"pow(x,25)", l_pow25, g_pow25, false}, {"pow(x,1/3.5)", l_powi, g_powi, false},
        {"atan2(x,.75)", l_atan2a, g_atan2a, false}, {"atan2(-1.25,x)", l_atan2b, g_atan2b, false}};
    for (auto& f : F) {
        std::atomic<long> bad{0};
        std::vector<std::thread> th;
        for (int t = 0; t < 8; ++t) th.emplace_back([&, t] {
            long b = 0;
            for (uint64_t u = (uint64_t)t << 29; u < ((uint64_t)(t + 1) << 29); ++u) {
                float x = gbl_asfloat((uint32_t)u);
                if (f.lim && !(fabsf(x) < 119.0f)) continue;
                float a = f.a(x), c = f.b(x);
                if (memcmp(&a, &c, 4) && !(a != a && c != c)) { if (b < 2) printf("  %s(%a): %a vs %a\n", f.name, x, a, c); ++b; }
            }
            bad += b;
        });
        for (auto& t : th) t.join();
        printf("%s: %ld mismatches over all floats\n", f.name, bad.load()); fflush(stdout);
    }
}
