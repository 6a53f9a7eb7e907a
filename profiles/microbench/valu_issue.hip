// Micro-benchmark behind DESIGN.md's VALU ceiling: how many wave64 VALU instructions one SIMD of an MI355X (gfx950) issues per
// second, for the instruction kinds the forward kernels are made of, at 1 / 2 / 4 / 8 resident waves per SIMD.
// Each wave runs ITER x 32 instructions over 8 independent registers (dependency distance 8 >= the pipeline depth), nothing
// else; blocks of 256 threads put one wave on each SIMD of a CU, W blocks per CU give W waves per SIMD.
// Round 3: every wave first waits at a counter in global memory until ALL waves of the launch have arrived, so that the W waves
// of a SIMD run their whole loop side by side (in round 2 the waves started as they were dispatched: a wave's own lifetime was
// shorter than the launch, and the two rates derived from them differed by 1.8x).  Reported per op and W: cycles (s_memtime) per
// instruction and SIMD, the shader clock those cycles ran at (cycles / 100 MHz wall clock), and three rates: from the waves' own
// timers (in_kernel), from the span first-wave-in .. last-wave-out of the loops on the device-wide 100 MHz counter (span: agrees with
// in_kernel when the waves really ran side by side), and from the HIP events around the launch (launch: includes the arrival
// counter — thousands of waves polling one address — so it stays below the other two).
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue > valu_issue.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int ITER = 40000;   // (long enough that the arrival counter — ~1 ms for 8192 waves on one address — is a few per cent of the launch)

#define REP8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define REP32(OP) REP8(OP) REP8(OP) REP8(OP) REP8(OP)

template <int KIND>
__global__ __launch_bounds__(256) void bench(uint32_t* out, unsigned long long* cyc, uint32_t k, uint32_t* arrived, uint32_t n_waves) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    // all waves of the launch resident and here before any starts its loop (bounded spin: a launch that does not fit would hang)
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(arrived, 1u);
        for (uint32_t spin = 0; spin < (1u << 22) && __hip_atomic_load(arrived, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n_waves; ++spin) __builtin_amdgcn_s_sleep(64);
    }
    __builtin_amdgcn_wave_barrier();
    const unsigned long long t0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < ITER; ++i) {
        if (KIND == 0) {
#define OP(r) asm volatile("v_pk_add_u16 %0, %0, %1" : "+v"(r) : "v"(k));
            REP32(OP)
#undef OP
        } else if (KIND == 1) {
#define OP(r) asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(r) : "v"(k));
            REP32(OP)
#undef OP
        } else if (KIND == 2) {
#define OP(r) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r) : "v"(k), "v"(0x05040100u));
            REP32(OP)
#undef OP
        } else if (KIND == 3) {
#define OP(r) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r));
            REP32(OP)
#undef OP
        } else if (KIND == 4) {
#define OP(r) asm volatile("v_pk_add_u16 %0, %0, %1 clamp" : "+v"(r) : "v"(k));
            REP32(OP)
#undef OP
        } else {
#define OP(r) asm volatile("v_add_u32 %0, %0, %1" : "+v"(r) : "v"(k));
            REP32(OP)
#undef OP
        }
    }
    const unsigned long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* c = cyc + 4 * (blockIdx.x * 4 + (threadIdx.x >> 6));
        c[0] = t1 - t0; c[1] = w1 - w0; c[2] = w0; c[3] = w1;   // (s_memrealtime is one counter for the whole device: begin / end are comparable across waves)
    }
}

template <int KIND>
int run(const char* name, int cus, uint32_t* d_out, unsigned long long* d_cyc, uint32_t* d_arr, bool first) {
    const int ws[4] = {1, 2, 4, 8};
    printf("%s  {\"op\": \"%s\", \"waves_per_simd\": {", first ? "" : ",\n", name);
    for (int wi = 0; wi < 4; ++wi) {
        const int W = ws[wi], blocks = cus * W;
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        CHECK(hipMemset(d_arr, 0, 4));
        hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 3u, d_arr, (uint32_t)blocks * 4);  // warm
        CHECK(hipDeviceSynchronize());
        CHECK(hipMemset(d_arr, 0, 4));
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(bench<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, d_cyc, 3u, d_arr, (uint32_t)blocks * 4);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> cyc(4 * (size_t)blocks * 4);
        CHECK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
        double c = 0, w = 0;
        unsigned long long first = ~0ull, last = 0;
        for (size_t i = 0; i < cyc.size(); i += 4) {
            c += (double)cyc[i]; w += (double)cyc[i + 1];
            if (cyc[i + 2] < first) first = cyc[i + 2];
            if (cyc[i + 3] > last) last = cyc[i + 3];
        }
        c /= (double)(cyc.size() / 4); w /= (double)(cyc.size() / 4);
        const double span_s = (double)(last - first) / 100e6;   // first wave into its loop .. last wave out of it
        const double instr_per_wave = (double)ITER * 32.0;
        // per SIMD: W waves x instr_per_wave instructions in the wave's lifetime (in-kernel clock) resp. the launch (events)
        const double wall_s = w / 100e6;  // wall_clock64: 100 MHz
        printf("%s\"%d\": {\"launch_ms\": %.4f, \"wave_cycles_clock64\": %.0f, \"wave_us_wallclock\": %.2f, "
               "\"simd_instr_per_s_in_kernel\": %.4g, \"simd_instr_per_s_launch\": %.4g, \"clock64_per_instr_per_wave\": %.3f, "
               "\"cycles_per_instr_per_simd\": %.3f, \"shader_clock_ghz\": %.3f, \"simd_instr_per_s_span\": %.4g, \"loops_span_ms\": %.4f}",
               wi ? ", " : "", W, ms, c, wall_s * 1e6, W * instr_per_wave / wall_s, W * instr_per_wave / (ms * 1e-3), c / instr_per_wave,
               c / (instr_per_wave * W), c / wall_s / 1e9, W * instr_per_wave / span_s, span_s * 1e3);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    }
    printf("}}");
    return 0;
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    int clk_khz = 0;
    (void)hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    uint32_t* d_out; unsigned long long* d_cyc; uint32_t* d_arr;
    CHECK(hipMalloc(&d_arr, 4));
    CHECK(hipMalloc(&d_out, (size_t)cus * 8 * 256 * 4));
    CHECK(hipMalloc(&d_cyc, (size_t)cus * 8 * 4 * 4 * 8));
    printf("{\"device_cus\": %d, \"clock_rate_khz\": %d, \"instr_per_wave\": %d, \"all_waves_start_together\": true, \"results\": [\n", cus, clk_khz, ITER * 32);
    if (run<0>("v_pk_add_u16", cus, d_out, d_cyc, d_arr, true)) return 1;
    if (run<4>("v_pk_add_u16 clamp", cus, d_out, d_cyc, d_arr, false)) return 1;
    if (run<1>("v_pk_min_u16", cus, d_out, d_cyc, d_arr, false)) return 1;
    if (run<2>("v_perm_b32", cus, d_out, d_cyc, d_arr, false)) return 1;
    if (run<3>("v_mov_b32_dpp row_shr:1", cus, d_out, d_cyc, d_arr, false)) return 1;
    if (run<5>("v_add_u32", cus, d_out, d_cyc, d_arr, false)) return 1;
    printf("\n]}\n");
    return 0;
}
