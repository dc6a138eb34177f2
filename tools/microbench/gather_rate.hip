// Vector-memory gather throughput per CU on gfx950 for the access shapes the SpMM kernels use.
// Build: hipcc -O3 --offload-arch=gfx950 -o gather_rate gather_rate.hip ; run: ./gather_rate
// Every wave issues ITER gather instructions (8 independent loads in flight), rows picked by an LCG from a table of
// `rows` rows; the sum of everything loaded is written out so nothing is dead.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ uint32_t lcg(uint32_t &s) { s = s * 1664525u + 1013904223u; return s >> 8; }

// mode 0: global_load_dwordx4, lanes_per_row lanes share a row (row_bytes = lanes_per_row*16)
// mode 1: same through a buffer descriptor (32-bit byte offsets)
// mode 2: global dwordx2 (row_bytes = lanes_per_row*8)
// mode 3: global dword
template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const float *__restrict__ table, uint32_t rows, int row_floats,
                                                int lanes_per_row, int active_lanes, int iters, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int g = lane / lanes_per_row, l = lane % lanes_per_row;
    uint32_t seed = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2654435761u + 12345u + g * 97u;   // same within a lane group
    f4 acc = {0, 0, 0, 0};
    // active_lanes < 0: the first -active_lanes lanes of EVERY lane group are on (rows narrower than their lane group)
    const bool on = active_lanes >= 0 ? lane < active_lanes : l < -active_lanes;
    auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)table, 0, 0xFFFFFFFFu, 0x00020000);
    for (int it = 0; it < iters; it += 8) {
        f4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t r = lcg(seed) % rows;
            v[j] = f4{0, 0, 0, 0};
            if (on) {
                if constexpr (MODE == 0) {
                    v[j] = *reinterpret_cast<const f4 *>(table + (size_t)r * row_floats + l * 4);
                } else if constexpr (MODE == 1) {
                    u4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (r * row_floats + l * 4) * 4, 0, 0);
                    v[j] = __builtin_bit_cast(f4, t);
                } else if constexpr (MODE == 2) {
                    f2 t = *reinterpret_cast<const f2 *>(table + (size_t)r * row_floats + l * 2);
                    v[j].x = t.x; v[j].y = t.y;
                } else {
                    v[j].x = table[(size_t)r * row_floats + l];
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc += v[j];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <int MODE>
double run(const float *table, uint32_t rows, int row_floats, int lanes_per_row, int active, int iters, float *out, int blocks) {
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, table, rows, row_floats, lanes_per_row, active, iters, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i)
        hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, table, rows, row_floats, lanes_per_row, active, iters, out);
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    return ms / 5 * 1e-3;
}

int main() {
    const size_t bytes = 512ull << 20;
    float *table, *out;
    CHECK(hipMalloc(&table, bytes)); CHECK(hipMemset(table, 0, bytes)); CHECK(hipMalloc(&out, 64));
    const int blocks = 256 * 8, iters = 512;   // 8 workgroups of 4 waves per CU
    const double instrs = (double)blocks * 4 * iters;
    struct Case { const char *name; int mode; uint32_t rows; int row_floats, lpr, active; };
    std::vector<Case> cases = {
        {"x4 4 rows x 256 B, 2 MB table (L2)", 0, 8192, 64, 16, 64},
        {"x4 4 rows x 256 B, 16 KB table (L1)", 0, 64, 64, 16, 64},
        {"x4 4 rows x 256 B, 400 MB table", 0, 1600000, 64, 16, 64},
        {"x4 buffer 4 rows x 256 B, 2 MB (L2)", 1, 8192, 64, 16, 64},
        {"x4 buffer 4 rows x 256 B, 16 KB (L1)", 1, 64, 64, 16, 64},
        {"x4 1 row x 1 KB, 2 MB (L2)", 0, 2048, 256, 64, 64},
        {"x4 1 row x 1 KB, 16 KB (L1)", 0, 16, 256, 64, 64},
        {"x4 2 rows x 512 B, 2 MB (L2)", 0, 4096, 128, 32, 64},
        {"x4 8 rows x 128 B, 2 MB (L2)", 0, 16384, 32, 8, 64},
        {"x4 16 rows x 64 B, 2 MB (L2)", 0, 32768, 16, 4, 64},
        {"x4 64 rows x 16 B, 2 MB (L2)", 0, 131072, 4, 1, 64},
        {"x4 4 rows x 256 B, 48 of 64 lanes on, L2", 0, 8192, 64, 16, 48},
        {"x4 4 rows x 256 B, 32 of 64 lanes on, L2", 0, 8192, 64, 16, 32},
        {"x4 4 rows x 256 B, 16 of 64 lanes on, L2", 0, 8192, 64, 16, 16},
        {"x2 2 rows x 256 B, 2 MB (L2)", 2, 8192, 64, 32, 64},
        {"x2 4 rows x 128 B, 2 MB (L2)", 2, 16384, 32, 16, 64},
        {"x1 1 row x 256 B, 2 MB (L2)", 3, 8192, 64, 64, 64},
        {"x1 64 lanes same 16 rows (4 lanes/row x 4 B)", 3, 8192, 64, 4, 64},
        // miss regimes: what does a gather instruction cost when its rows are 2 lines, 1 line, half a line?
        {"x4 4 rows x 256 B, 14 MB table (user step)", 0, 54571, 64, 16, 64},
        {"x4 4 rows x 128 B of 256-B rows, 14 MB", 0, 54571, 64, 16, -8},
        {"x4 4 rows x 64 B of 256-B rows, 14 MB", 0, 54571, 64, 16, -4},
        {"x4 8 rows x 128 B, 14 MB table", 0, 109142, 32, 8, 64},
        {"x4 4 rows x 128 B of 256-B rows, 400 MB", 0, 1600000, 64, 16, -8},
        {"x4 4 rows x 64 B of 256-B rows, 400 MB", 0, 1600000, 64, 16, -4},
        {"x4 8 rows x 128 B, 400 MB table", 0, 3200000, 32, 8, 64},
        {"x4 2 rows x 512 B, 400 MB table", 0, 800000, 128, 32, 64},
        {"x4 2 rows x 384 B of 512-B rows, 400 MB", 0, 800000, 128, 32, -24},
    };
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const double clk = prop.clockRate * 1e3;
    printf("device %s, %d CUs, clock %.0f MHz\n", prop.name, prop.multiProcessorCount, clk / 1e6);
    for (auto &c : cases) {
        double t;
        if (c.mode == 0) t = run<0>(table, c.rows, c.row_floats, c.lpr, c.active, iters, out, blocks);
        else if (c.mode == 1) t = run<1>(table, c.rows, c.row_floats, c.lpr, c.active, iters, out, blocks);
        else if (c.mode == 2) t = run<2>(table, c.rows, c.row_floats, c.lpr, c.active, iters, out, blocks);
        else t = run<3>(table, c.rows, c.row_floats, c.lpr, c.active, iters, out, blocks);
        const double per_lane = c.mode <= 1 ? 16 : c.mode == 2 ? 8 : 4;
        const int n_active = c.active >= 0 ? c.active : (64 / c.lpr) * -c.active;
        const double useful = instrs * n_active * per_lane;
        printf("%-50s %8.1f us  %6.2f TB/s useful  %5.1f ns/instr/CU = %5.1f clk@2.1GHz  %5.1f B/clk/CU\n", c.name, t * 1e6,
               useful / t / 1e12, t / (instrs / 256) * 1e9, t / (instrs / 256) * 2.1e9, useful / t / 256 / 2.1e9);
    }
    return 0;
}
