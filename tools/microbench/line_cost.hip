// What a 128-B line costs a CU, by kind, and whether the kinds overlap -- the cost model of DESIGN.md section 5.
// Build: hipcc -O3 --offload-arch=gfx950 -o line_cost line_cost.hip ; run: ./line_cost
// Every wave runs ITER rounds of G gather instructions (4 random 256-B rows each, 8 in flight) from a table of
// `rows` rows and, every round, S store instructions (4 random 256-B rows of a 420 MB output).  Cases: gathers alone
// (L2-resident / Infinity-Cache-resident / HBM-resident table), stores alone (plain, nt, sc1, sc0 sc1), and both --
// against the sum of the two alone ("sum") and the larger of them ("max").
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int SAUX>
__global__ __launch_bounds__(256) void k_mix(const float *__restrict__ table, uint32_t rows, float *__restrict__ out,
                                             uint32_t out_rows, int iters, int gathers, int stores, float *__restrict__ sink) {
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, l = lane & 15;
    uint32_t seed = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u + g * 97u;
    f4 acc = {0, 0, 0, 0};
    auto tsrc = __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, 0xFFFFFFFFu, 0x00020000);
    auto osrc = __builtin_amdgcn_make_buffer_rsrc((void *)out, 0, 0xFFFFFFFFu, 0x00020000);
    for (int it = 0; it < iters; ++it) {
        for (int j0 = 0; j0 < gathers; j0 += 8) {
            f4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t r = lcg(seed) % rows;
                v[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(tsrc, (r * 64 + l * 4) * 4, 0, 0));
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
        for (int j = 0; j < stores; ++j) {
            const uint32_t r = lcg(seed) % out_rows;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, acc), osrc, (r * 64 + l * 4) * 4, 0, SAUX);
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) sink[0] = acc.x;
}

template <int SAUX>
double run(const float *table, uint32_t rows, float *out, uint32_t out_rows, int iters, int gathers, int stores, float *sink) {
    const int blocks = 256 * 8;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_mix<SAUX>, dim3(blocks), dim3(256), 0, 0, table, rows, out, out_rows, iters, gathers, stores, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 3; ++i)
        hipLaunchKernelGGL(k_mix<SAUX>, dim3(blocks), dim3(256), 0, 0, table, rows, out, out_rows, iters, gathers, stores, sink);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 3 * 1e-3;
}

double run_aux(int aux, const float *table, uint32_t rows, float *out, uint32_t out_rows, int iters, int gathers, int stores, float *sink) {
    switch (aux) {
        case 0: return run<0>(table, rows, out, out_rows, iters, gathers, stores, sink);
        case 2: return run<2>(table, rows, out, out_rows, iters, gathers, stores, sink);
        case 16: return run<16>(table, rows, out, out_rows, iters, gathers, stores, sink);
        case 17: return run<17>(table, rows, out, out_rows, iters, gathers, stores, sink);
        default: return run<18>(table, rows, out, out_rows, iters, gathers, stores, sink);
    }
}

int main() {
    const size_t tbytes = 512ull << 20, obytes = 420ull << 20;
    float *table, *out, *sink;
    CHECK(hipMalloc(&table, tbytes)); CHECK(hipMemset(table, 0, tbytes));
    CHECK(hipMalloc(&out, obytes)); CHECK(hipMemset(out, 0, obytes));
    CHECK(hipMalloc(&sink, 64));
    const uint32_t out_rows = (uint32_t)(obytes / 256);
    const int iters = 40;
    const double waves = 256.0 * 8 * 4;
    struct T { const char *name; uint32_t rows; };
    const T tables[] = {{"2 MB (L2)", 8192}, {"14 MB (Infinity Cache)", 54571}, {"400 MB (HBM)", 1600000}};
    const int auxes[] = {0, 2, 16, 17, 18};
    const char *aux_name[] = {"plain", "nt", "sc1", "sc0 sc1", "nt sc1"};
    printf("per CU: ns per 128-B line (a gather or store instruction moves 8 lines)\n");
    double t_store[5];
    for (int a = 0; a < 5; ++a) {
        t_store[a] = run_aux(auxes[a], table, 8192, out, out_rows, iters, 0, 8, sink);
        const double lines = waves * iters * 8 * 8;
        printf("stores alone, %-8s                      %8.1f us   %5.2f ns/line/CU   %5.2f TB/s\n", aux_name[a], t_store[a] * 1e6,
               t_store[a] / (lines / 256) * 1e9, lines * 128 / t_store[a] / 1e12);
    }
    for (const T &t : tables) {
        const double tg = run_aux(0, table, t.rows, out, out_rows, iters, 48, 0, sink);
        const double glines = waves * iters * 48 * 8;
        printf("gathers alone, %-24s       %8.1f us   %5.2f ns/line/CU   %5.2f TB/s\n", t.name, tg * 1e6, tg / (glines / 256) * 1e9,
               glines * 128 / tg / 1e12);
        for (int a = 0; a < 5; ++a) {
            const double tm = run_aux(auxes[a], table, t.rows, out, out_rows, iters, 48, 8, sink);
            printf("  + 8 stores per 48 gathers, %-8s        %8.1f us   sum %8.1f  max %8.1f   -> %4.0f %% of the way from max to sum\n",
                   aux_name[a], tm * 1e6, (tg + t_store[a]) * 1e6, (tg > t_store[a] ? tg : t_store[a]) * 1e6,
                   100.0 * (tm - (tg > t_store[a] ? tg : t_store[a])) / ((tg < t_store[a] ? tg : t_store[a]) + 1e-12));
        }
    }
    // hits and misses mixed: half the gathers from the 2 MB table, half from the 400 MB one
    return 0;
}
