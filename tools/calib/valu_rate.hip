// VALU issue-rate probe for gfx950: how many wave64 vector instructions per cycle one SIMD retires, for the
// instruction kinds the traversal step is made of, at 1..8 resident waves per SIMD.  Answers two questions the
// what-if table in DESIGN.md leaves open: (1) is v_pk_fma_f32 (two f32 FMAs per instruction, bit-identical to two
// v_fma_f32) full rate, i.e. would packing the 12 quotient refinements of the slab test halve their issue slots;
// (2) how much of the SIMD's VALU issue does a dependent chain of one wave use.
// Build + run: hipcc -O3 --offload-arch=gfx950 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kIters = 16384;      // loop trips
constexpr int kPerTrip = 32;      // instructions per trip

// KIND 0: v_fma_f32, 8 independent accumulators; 1: v_fma_f32, ONE dependent chain; 2: v_pk_fma_f32, 8 independent pairs;
// 3: v_pk_fma_f32 one dependent chain; 4: v_rcp_f32 independent; 5: v_min3_f32 independent; 6: v_fma_f64 independent;
// 7: v_pk_mul_f32 independent; 8: v_pk_add_f32 independent; 9: v_cndmask_b32 independent
template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(float *out, float seed) {
    float a[8]; f2 p[8]; double d[8];
    const float x = seed + (float)threadIdx.x * 1e-9f, y = 0.999999f;
    const f2 x2 = {x, x}, y2 = {y, y};
    const double xd = x, yd = y;
    unsigned long long mask = __ballot(threadIdx.x & 1);
    unsigned sc = 0;
    if (KIND == 32 || KIND == 42) asm volatile("s_mov_b64 vcc, %0" : : "s"(mask) : "vcc");
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = x + i; p[i] = (f2){x + i, x - i}; d[i] = xd + i; }
    for (int it = 0; it < kIters; it++) {
#pragma unroll
        for (int k = 0; k < kPerTrip; k++) {
            const int i = k & 7;
            if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y));
            if (KIND == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a[0]) : "v"(x), "v"(y));
            if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[i]) : "v"(x2), "v"(y2));
            if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[0]) : "v"(x2), "v"(y2));
            if (KIND == 4) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 5) asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            if (KIND == 6) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(xd), "v"(yd));
            if (KIND == 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(y2));
            if (KIND == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(y2));
            if (KIND == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
            if (KIND == 10) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "s"(mask));
            if (KIND == 11) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(x));
            if (KIND == 12) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
            if (KIND == 13) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
            if (KIND == 14) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
            if (KIND == 15) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
            if (KIND == 16) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(d[i]) : "v"(x), "v"(y) : "vcc");
            if (KIND == 17) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(y) : "vcc");
            if (KIND == 18) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            if (KIND == 19) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            if (KIND == 20) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(yd));
            if (KIND == 21) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(x) : "vcc");
            if (KIND == 22) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
            if (KIND == 23) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(d[i]) : "v"(xd));
            if (KIND == 24) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]));
            if (KIND == 25) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            if (KIND == 26) asm volatile("v_mov_b64 %0, %1" : "+v"(d[i]) : "v"(xd));
            if (KIND == 27) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 28) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(yd));
            if (KIND == 31) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
            if (KIND == 32) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));      // vcc set by s_mov before the loop
            if (KIND == 33) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x));
            if (KIND == 34) asm volatile("v_min_u32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
            if (KIND == 35) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            if (KIND == 36) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(y));
            if (KIND == 37) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y));
            if (KIND == 38) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
            if (KIND == 39) asm volatile("v_lshlrev_b32 %0, 3, %0" : "+v"(a[i]));
            if (KIND == 40) asm volatile("v_max_f32_e64 %0, |%0|, %1" : "+v"(a[i]) : "v"(x));
            if (KIND == 41) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(mask) : "v"(a[i]), "v"(x));
            if (KIND == 42) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(y));        // other operand: constant-ish y
            if (KIND == 43) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(a[i]));
            if (KIND == 44) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(a[i]) : "v"(x) : "vcc");
            if (KIND == 45) asm volatile("v_readfirstlane_b32 %0, %1" : "=s"(sc) : "v"(a[i]));
            if (KIND == 29) asm volatile("s_and_b64 %0, %0, exec" : "+s"(mask) : : "scc");
            if (KIND == 30) asm volatile("v_fma_f32 %0, %2, %3, %0\n\ts_and_b64 %1, %1, exec" : "+v"(a[i]), "+s"(mask) : "v"(x), "v"(y) : "scc");
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)d[i];
    if (s == 123.456f || mask == 12345ull || sc == 77u) out[0] = s;
}

template <int KIND>
static void run(const char *name, int waves_per_simd, float *out) {
    // 256-thread blocks = 4 waves = one per SIMD of a CU; waves_per_simd blocks per CU
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int grid = cus * waves_per_simd;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    rate_kernel<KIND><<<grid, 256>>>(out, 1.0f);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    rate_kernel<KIND><<<grid, 256>>>(out, 1.0f);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double insts_per_simd = (double)kIters * kPerTrip * waves_per_simd * ((KIND == 30 || KIND == 31) ? 1 : 1);     // wave-instructions each SIMD retired
    const double mhz = prop.clockRate / 1000.0;                                   // nominal peak engine clock
    printf("{\"kind\": \"%s\", \"waves_per_simd\": %d, \"ms\": %.4f, \"wave_insts_per_simd_per_us\": %.1f, \"cycles_per_inst_at_%.0fMHz\": %.3f}\n",
           name, waves_per_simd, ms, insts_per_simd / (ms * 1e3), mhz, (ms * 1e-3) * mhz * 1e6 / insts_per_simd);
    fflush(stdout);
}

int main(int argc, char **argv) {
    float *out; CHECK(hipMalloc(&out, 64));
    const int ws[] = {1, 5, 8};
    for (int w : ws) {
        if (argc > 1 && atoi(argv[1]) != w) continue;
        run<0>("v_fma_f32 independent", w, out);
        run<1>("v_fma_f32 dependent chain", w, out);
        run<2>("v_pk_fma_f32 independent", w, out);
        run<3>("v_pk_fma_f32 dependent chain", w, out);
        run<4>("v_rcp_f32", w, out);
        run<5>("v_min3_f32", w, out);
        run<6>("v_fma_f64", w, out);
        run<7>("v_pk_mul_f32", w, out);
        run<8>("v_pk_add_f32", w, out);
        run<9>("v_cndmask_b32 vcc", w, out);
        run<10>("v_cndmask_b32 sgpr mask", w, out);
        run<11>("v_mov_b32", w, out);
        run<12>("v_max_f32", w, out);
        run<13>("v_mul_f32", w, out);
        run<14>("v_add_u32", w, out);
        run<15>("v_mul_lo_u32", w, out);
        run<16>("v_mad_u64_u32", w, out);
        run<17>("v_div_scale_f32", w, out);
        run<18>("v_div_fmas_f32", w, out);
        run<19>("v_div_fixup_f32", w, out);
        run<20>("v_mul_f64", w, out);
        run<21>("v_cmp_lt_f32 -> vcc", w, out);
        run<22>("v_xor_b32", w, out);
        run<23>("v_lshl_add_u64", w, out);
        run<24>("v_cvt_f32_u32", w, out);
        run<25>("v_max3_f32", w, out);
        run<26>("v_mov_b64", w, out);
        run<27>("v_sqrt_f32", w, out);
        run<28>("v_add_f64", w, out);
        run<31>("v_cmp_lt_f32 vcc + v_cndmask_b32 vcc (pair = 2 insts)", w, out);
        run<32>("v_cndmask_b32 vcc (vcc written by SALU)", w, out);
        run<33>("v_cndmask_b32_e64 vcc", w, out);
        run<34>("v_min_u32", w, out);
        run<35>("v_med3_f32", w, out);
        run<36>("v_sub_f32", w, out);
        run<37>("v_fmac_f32", w, out);
        run<38>("v_and_b32", w, out);
        run<39>("v_lshlrev_b32", w, out);
        run<40>("v_max_f32_e64 |abs|", w, out);
        run<41>("v_cmp_lt_f32_e64 -> sgpr", w, out);
        run<42>("v_cndmask_b32 vcc, non-lane-varying source", w, out);
        run<43>("v_bfe_u32", w, out);
        run<44>("v_add_co_u32", w, out);
        run<45>("v_readfirstlane_b32", w, out);
        run<29>("s_and_b64 (SALU only)", w, out);
        run<30>("v_fma_f32 + s_and_b64 pairs", w, out);
    }
    CHECK(hipFree(out));
    return 0;
}
