// lgconv_hip.hip -- LightGCN propagation for MI355X (gfx950, CDNA4).  C ABI: include/lgconv_hip.h
//
// Memory-bound sparse aggregation (<= 2.5 FLOP/B): no MFMA, no LDS tiling of operands.
// What matters here is (1) every embedding row moving as full-width, coalesced vector
// loads/stores -- one *lane group* of LPR = ceil(D/4) lanes owns a row and each lane a
// float4 of it, so a 64-wide wavefront carries 64/LPR rows per instruction (4 at D=64);
// (2) many independent row gathers in flight per wave (4-way unrolled entry loop, 8 waves
// per SIMD at <= 64 VGPRs); (3) no float atomics on the forward path: long rows are cut into
// chunks whose partial sums are combined in a fixed order.
//
// Kernels
//   k_build_*        one-time COO -> CSR (+ reference-order fp32 degree, normalised values)
//   k_spmm_hop       one launch per operator and hop:
//       chunk part   one wavefront per chunk of a long row, lane groups stride the entries,
//                    cross-group reduce through ds_bpermute shuffles
//       row part     rows with <= short_max entries, one lane group per row, entry order, entries
//                    fetched through a fixed-width slab of row heads
//   k_spmm_combine   fixed-order sum of a long row's partial slots + epilogue
//   k_lincomb, k_pair_dot, k_pair_dot_bwd
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <chrono>
#include <new>
#include <thread>
#include <vector>

#include "lgconv_hip.h"

namespace {

constexpr int kWave = 64;
constexpr int kBlock = 256;  // 4 wavefronts, one per SIMD of a CU

typedef float f4 __attribute__((ext_vector_type(4)));
// Row slices are only dword-aligned in general (D = 90 -> 360-byte rows; the overlapping last lane):
// tell the compiler, it still emits global_load_dwordx4 (gfx950 allows dword-aligned wide accesses).
typedef f4 f4u __attribute__((aligned(4)));

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }
inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// ----------------------------------------------------------------------------------------
// Graph build
// ----------------------------------------------------------------------------------------
__global__ void k_build_keys(const int64_t *__restrict__ edge_index, int64_t n_edges, int64_t n_nodes,
                             int by_source, int32_t *__restrict__ keys, int32_t *__restrict__ ids,
                             int32_t *__restrict__ status) {
    int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    int64_t src = edge_index[e], dst = edge_index[n_edges + e];
    bool ok = src >= 0 && src < n_nodes && dst >= 0 && dst < n_nodes;
    if (!ok) atomicOr(status, LGC_ST_INDEX_OOB);
    // an invalid edge is parked on row 0 and later given value 0 / column 0
    keys[e] = ok ? (int32_t)(by_source ? src : dst) : 0;
    ids[e] = (int32_t)e;
}

__global__ void k_build_rowptr(const int32_t *__restrict__ sorted_keys, int32_t n_edges, int32_t n_nodes,
                               int32_t *__restrict__ rowptr) {
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r > n_nodes) return;
    int32_t lo = 0, hi = n_edges;  // first position whose key >= r
    while (lo < hi) {
        int32_t mid = lo + ((hi - lo) >> 1);
        if (sorted_keys[mid] < r) lo = mid + 1; else hi = mid;
    }
    rowptr[r] = lo;
}

__global__ void k_build_sorted_weight(const int32_t *__restrict__ perm, const float *__restrict__ w,
                                      int32_t n_edges, float *__restrict__ w_sorted) {
    int32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n_edges) w_sorted[k] = w ? w[perm[k]] : 1.0f;
}

constexpr int kDegSerialMax = 512;

// deg[r] = ((0 + w_0) + w_1) + ...  in edge order: bitwise what a sequential CPU scatter_add gives.
__global__ void k_build_degree_serial(const int32_t *__restrict__ rowptr, const float *__restrict__ w_sorted,
                                      int32_t n_nodes, float *__restrict__ deg) {
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_nodes) return;
    int32_t s = rowptr[r], e = rowptr[r + 1];
    if (e - s > kDegSerialMax) return;
    float acc = 0.0f;
    for (int32_t k = s; k < e; ++k) acc = __fadd_rn(acc, w_sorted[k]);
    deg[r] = acc;
}

// Hub rows: a wavefront streams the row 64 values at a time (coalesced) and lane order IS edge
// order, so the serial chain runs over v_readlane without touching memory again.
__global__ void k_build_degree_wave(const int32_t *__restrict__ rowptr, const float *__restrict__ w_sorted,
                                    int32_t n_nodes, float *__restrict__ deg) {
    int32_t r = blockIdx.x * (blockDim.x / kWave) + (threadIdx.x / kWave);
    if (r >= n_nodes) return;
    int32_t s = rowptr[r], e = rowptr[r + 1];
    if (e - s <= kDegSerialMax) return;
    const int lane = threadIdx.x & (kWave - 1);
    float acc = 0.0f;
    for (int32_t base = s; base < e; base += kWave) {
        int32_t k = base + lane;
        float v = k < e ? w_sorted[k] : 0.0f;
        int n = min(kWave, e - base);
        if (n == kWave) {
#pragma unroll
            for (int i = 0; i < kWave; ++i) acc = __fadd_rn(acc, __shfl(v, i));
        } else {
            for (int i = 0; i < n; ++i) acc = __fadd_rn(acc, __shfl(v, i));
        }
    }
    if (lane == 0) deg[r] = acc;
}

__global__ void k_build_dis(const float *__restrict__ deg, int32_t n_nodes, float *__restrict__ dis) {
    int32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_nodes) return;
    // deg^-1/2 as 1/sqrt with both steps correctly rounded; +inf (deg == 0) -> 0, NaN (deg < 0) stays.
    float d = 1.0f / sqrtf(deg[r]);  // hipcc default: both correctly rounded (NOT __fsqrt_rn = bare v_sqrt_f32, 1 ulp)
    dis[r] = (d == INFINITY) ? 0.0f : d;
}

__global__ void k_build_entries(const int64_t *__restrict__ edge_index, const float *__restrict__ w,
                                const int32_t *__restrict__ perm, const float *__restrict__ dis,
                                int64_t n_edges, int64_t n_nodes, int by_source, int normalize,
                                lgc_entry *__restrict__ entries, float *__restrict__ edge_val) {
    int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_edges) return;
    int32_t e = perm[k];
    int64_t src = edge_index[e], dst = edge_index[n_edges + e];
    bool ok = src >= 0 && src < n_nodes && dst >= 0 && dst < n_nodes;
    float wv = w ? w[e] : 1.0f;
    float v = 0.0f;
    int32_t c = 0;
    if (ok) {
        // left to right, each product rounded: dis[j] * w * dis[i]
        v = normalize ? __fmul_rn(__fmul_rn(dis[src], wv), dis[dst]) : wv;
        c = (int32_t)(by_source ? dst : src);
    }
    entries[k].col = c;
    entries[k].val = v;
    if (edge_val) edge_val[e] = v;
}

struct BuildWs {
    int32_t *keys_in, *keys_out, *ids_in, *ids_out;
    float *w_sorted;
    void *cub;
    size_t cub_bytes;
    size_t total;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

int end_bit_for(int64_t n_nodes) {
    int bits = 1;
    while (bits < 31 && (int64_t(1) << bits) < n_nodes) ++bits;
    return bits;
}

BuildWs carve_build_ws(void *base, int64_t n_nodes, int64_t n_edges) {
    BuildWs ws{};
    size_t cub_bytes = 0;
    int32_t *nul = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, cub_bytes, nul, nul, nul, nul, (int)n_edges, 0,
                                             end_bit_for(n_nodes), (hipStream_t)0);
    size_t arr = align_up((size_t)n_edges * 4, 256);
    const uintptr_t p = reinterpret_cast<uintptr_t>(base);   // integer arithmetic: base is null when only sizing
    ws.keys_in = reinterpret_cast<int32_t *>(p);
    ws.keys_out = reinterpret_cast<int32_t *>(p + arr);
    ws.ids_in = reinterpret_cast<int32_t *>(p + 2 * arr);
    ws.ids_out = reinterpret_cast<int32_t *>(p + 3 * arr);
    ws.w_sorted = reinterpret_cast<float *>(p);  // reuses keys_in after the sort
    ws.cub = reinterpret_cast<void *>(p + 4 * arr);
    ws.cub_bytes = align_up(cub_bytes, 256);
    ws.total = 4 * arr + ws.cub_bytes + 256;
    return ws;
}

// ----------------------------------------------------------------------------------------
// SpMM
// ----------------------------------------------------------------------------------------
struct SpmmArgs {
    const int32_t *rowptr;
    const lgc_entry *entries;
    const float *x;
    float *y;
    const float *r;
    int64_t x_stride, y_stride, r_stride;
    float a, b;
    int32_t dim, lpr;  // lanes per row = ceil(dim / VEC)
    int32_t row_begin, row_end, short_max;
    int32_t wt_store;  // 1: output rows leave with write-through (sc1) stores, see store_out
    // seeded pull (lgc_seed_pull): only columns with col_flag != 0 are gathered, from row col_slot[col] of x
    const uint8_t *col_flag;
    const int32_t *col_slot;
    const uint8_t *row_mark;   // optional: rows with row_mark[row] == 0 have no flagged column and are written as zeros unread
};

// A lane's slice of a row: VEC consecutive floats starting at column c0.
template <int VEC>
struct Acc {
    float v[VEC];
    __device__ __forceinline__ void zero() {
#pragma unroll
        for (int i = 0; i < VEC; ++i) v[i] = 0.0f;
    }
};

// Column of lane l's slice.  When D is not a multiple of 4 (D = 90) the LAST lane of the group takes
// the final four columns, overlapping its neighbour by 4*LPR - D columns: every lane then issues the same
// dwordx4 (no divergent narrow access for a tail), both lanes compute identical values for the shared
// columns, and the overlapping stores write the same bits.
template <int VEC>
__device__ __forceinline__ int lane_column(int l, int dim) {
    return VEC == 1 ? l : min(l * VEC, dim - VEC);
}

template <int VEC>
__device__ __forceinline__ Acc<VEC> load_row(const float *p) {
    Acc<VEC> o;
    if constexpr (VEC == 4) {
        f4 t = *reinterpret_cast<const f4u *>(p);
        o.v[0] = t.x; o.v[1] = t.y; o.v[2] = t.z; o.v[3] = t.w;
    } else {
        o.v[0] = p[0];
    }
    return o;
}

// The table row an entry gathers.  FILTER (lgc_seed_pull): x is a compact table of seed rows, an entry counts only if
// its column carries a flag and then reads the compact row col_slot[col]; everything else contributes nothing.
template <int VEC, bool FILTER, class P>
__device__ __forceinline__ Acc<VEC> gather_row(const P &p, const float *xb, int32_t col) {
    if constexpr (!FILTER) {
        return load_row<VEC>(xb + (int64_t)col * p.x_stride);
    } else {
        Acc<VEC> o;
        o.zero();
        if (p.col_flag[col]) o = load_row<VEC>(xb + (int64_t)p.col_slot[col] * p.x_stride);
        return o;
    }
}

template <int VEC>
__device__ __forceinline__ void store_row(float *p, const Acc<VEC> &o) {
    if constexpr (VEC == 4) {
        f4 t = {o.v[0], o.v[1], o.v[2], o.v[3]};
        *reinterpret_cast<f4u *>(p) = t;
    } else {
        p[0] = o.v[0];
    }
}

// Output rows are written once and not read again in this launch.  A plain store keeps its line in the
// XCD's 4 MiB L2, where 0.42 GB of output per layer evicts the gathered rows (the only data with reuse);
// an sc1 (write-through) store drops the line from L2 instead (MI355X_MICROARCH.md, table of store
// flavours).  Needs a buffer descriptor: 32-bit byte offsets, so only for tables below 4 GiB (wt_store).
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
template <int VEC, class P>
__device__ __forceinline__ void store_out(const P &p, int64_t row, int c0, const Acc<VEC> &o) {
    if constexpr (VEC == 4) {
        if (p.wt_store) {
            auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, 0xFFFFFFFFu, 0x00020000);
            f4 t = {o.v[0], o.v[1], o.v[2], o.v[3]};
            const unsigned off = (unsigned)((row * p.y_stride + c0) * 4);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, t), rsrc, off, 0, /*sc1*/ 16);
            return;
        }
    }
    store_row<VEC>(p.y + row * p.y_stride + c0, o);
}

// acc = acc + val * x, product rounded before the add: the arithmetic of the reference's
// "scale the gathered rows, then index_add_" (no FMA contraction; file built with -ffp-contract=off).
template <int VEC>
__device__ __forceinline__ void mul_add(Acc<VEC> &acc, float val, const Acc<VEC> &x) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.v[i] = __fadd_rn(acc.v[i], __fmul_rn(val, x.v[i]));
}

// y = a * acc + b * rv
template <int VEC, class P>
__device__ __forceinline__ void finish_row(const P &p, int64_t row, int c0, Acc<VEC> acc, const Acc<VEC> &rv) {
    if (p.a != 1.0f) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc.v[i] = __fmul_rn(p.a, acc.v[i]);
    }
    if (p.r != nullptr) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc.v[i] = __fadd_rn(acc.v[i], __fmul_rn(p.b, rv.v[i]));
    }
    store_out<VEC, P>(p, row, c0, acc);
}

// Short rows: one lane group per row, entries in order, 4 gathers in flight per group; the epilogue
// row is requested before the gathers and consumed after them.
template <int VEC, bool FILTER = false>
__device__ __forceinline__ void rows_body(const SpmmArgs &p, int64_t block) {
    const int lane = threadIdx.x & (kWave - 1);
    const int rows_per_wave = kWave / p.lpr;
    const int g = lane / p.lpr;
    const int l = lane - g * p.lpr;
    const int64_t wave = block * (kBlock / kWave) + (threadIdx.x / kWave);
    const int64_t row = (int64_t)p.row_begin + wave * rows_per_wave + g;
    if (g >= rows_per_wave || row >= p.row_end) return;
    const int c0 = lane_column<VEC>(l, p.dim);

    const int32_t s = p.rowptr[row], e = p.rowptr[row + 1];
    if (e - s > p.short_max) return;  // belongs to the chunk part
    if constexpr (FILTER) {
        if (p.row_mark != nullptr && p.row_mark[row] == 0) {   // no seed among this row's columns (lgc_seed_mark)
            Acc<VEC> zero;
            zero.zero();
            store_out<VEC, SpmmArgs>(p, row, c0, zero);
            return;
        }
    }

    const float *xb = p.x + c0;
    const lgc_entry *__restrict__ ent = p.entries;
    Acc<VEC> acc, rv;
    acc.zero();
    rv.zero();
    if (p.r != nullptr) rv = load_row<VEC>(p.r + row * p.r_stride + c0);
    int32_t k = s;
    for (; k + 4 <= e; k += 4) {
        lgc_entry e0 = ent[k], e1 = ent[k + 1], e2 = ent[k + 2], e3 = ent[k + 3];
        Acc<VEC> x0 = gather_row<VEC, FILTER>(p, xb, e0.col);
        Acc<VEC> x1 = gather_row<VEC, FILTER>(p, xb, e1.col);
        Acc<VEC> x2 = gather_row<VEC, FILTER>(p, xb, e2.col);
        Acc<VEC> x3 = gather_row<VEC, FILTER>(p, xb, e3.col);
        mul_add<VEC>(acc, e0.val, x0);
        mul_add<VEC>(acc, e1.val, x1);
        mul_add<VEC>(acc, e2.val, x2);
        mul_add<VEC>(acc, e3.val, x3);
    }
    for (; k < e; ++k) {
        lgc_entry e0 = ent[k];
        Acc<VEC> x0 = gather_row<VEC, FILTER>(p, xb, e0.col);
        mul_add<VEC>(acc, e0.val, x0);
    }
    finish_row<VEC, SpmmArgs>(p, row, c0, acc, rv);
}

// Long rows: one wavefront per chunk.  Lane group g takes entries begin+g, begin+g+G, ...;
// the G group sums are then added in group order by lane group 0.
template <int VEC, bool FILTER = false, bool SCREEN = false>
__device__ __forceinline__ void chunk_core(const SpmmArgs &p, const lgc_chunk ch, float *__restrict__ partials) {
    const int lane = threadIdx.x & (kWave - 1);
    const int groups = kWave / p.lpr;
    const int g = lane / p.lpr;
    const int l = lane - g * p.lpr;
    const int c0 = lane_column<VEC>(l, p.dim);
    const bool active = g < groups;

    Acc<VEC> acc;
    acc.zero();
    if constexpr (SCREEN) {
        // a marked chunk of a seeded pull: 64 entries and their flags per round trip, then only the flagged entries (a few
        // per chunk) are gathered -- by the lane group that owns them in the plain order (entry k belongs to group
        // (k - begin) % groups, groups are added up in group order below), so the sums have the bits of the full scan:
        // an unflagged entry adds val * 0 there
        const lgc_entry *__restrict__ ent = p.entries;
        for (int32_t base = ch.begin; base < ch.end; base += kWave) {
            const int32_t k = base + lane;
            lgc_entry e = {0, 0.0f};
            bool f = false;
            if (k < ch.end) {
                e = ent[k];
                f = p.col_flag[e.col] != 0;
            }
            unsigned long long m = __ballot(f);
            while (m != 0) {   // wave-uniform
                const int j = __builtin_ctzll(m);
                m &= m - 1;
                const int32_t col = __shfl(e.col, j);
                const float val = __shfl(e.val, j);
                if (active && g == (base + j - ch.begin) % groups) {
                    const Acc<VEC> x = load_row<VEC>(p.x + c0 + (int64_t)p.col_slot[col] * p.x_stride);
                    mul_add<VEC>(acc, val, x);
                }
            }
        }
    } else if (active) {
        const float *xb = p.x + c0;
        const lgc_entry *__restrict__ ent = p.entries;
        int32_t k = ch.begin + g;
        // software pipeline: the entries of step t+1 are requested before the gathers of step t are
        // consumed, so a step costs one memory round trip (the gathers) instead of two
        const int32_t step = 4 * groups;
        lgc_entry n0, n1, n2, n3;
        bool more = k + 3 * groups < ch.end;
        if (more) { n0 = ent[k]; n1 = ent[k + groups]; n2 = ent[k + 2 * groups]; n3 = ent[k + 3 * groups]; }
        while (more) {
            const lgc_entry e0 = n0, e1 = n1, e2 = n2, e3 = n3;
            Acc<VEC> x0 = gather_row<VEC, FILTER>(p, xb, e0.col);
            Acc<VEC> x1 = gather_row<VEC, FILTER>(p, xb, e1.col);
            Acc<VEC> x2 = gather_row<VEC, FILTER>(p, xb, e2.col);
            Acc<VEC> x3 = gather_row<VEC, FILTER>(p, xb, e3.col);
            k += step;
            more = k + 3 * groups < ch.end;
            if (more) { n0 = ent[k]; n1 = ent[k + groups]; n2 = ent[k + 2 * groups]; n3 = ent[k + 3 * groups]; }
            mul_add<VEC>(acc, e0.val, x0);
            mul_add<VEC>(acc, e1.val, x1);
            mul_add<VEC>(acc, e2.val, x2);
            mul_add<VEC>(acc, e3.val, x3);
        }
        for (; k < ch.end; k += groups) {
            lgc_entry e0 = ent[k];
            Acc<VEC> x0 = gather_row<VEC, FILTER>(p, xb, e0.col);
            mul_add<VEC>(acc, e0.val, x0);
        }
    }
    // every lane executes the shuffles; only group 0 keeps the result
    for (int j = 1; j < groups; ++j) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float other = __shfl_down(acc.v[i], j * p.lpr);
            if (g == 0) acc.v[i] = __fadd_rn(acc.v[i], other);
        }
    }
    if (!active || g != 0) return;
    if (ch.slot >= 0) {
        store_row<VEC>(partials + (int64_t)ch.slot * p.dim + c0, acc);
    } else {
        Acc<VEC> rv;
        rv.zero();
        if (p.r != nullptr) rv = load_row<VEC>(p.r + (int64_t)ch.row * p.r_stride + c0);
        finish_row<VEC, SpmmArgs>(p, ch.row, c0, acc, rv);
    }
}

template <int VEC, bool FILTER = false>
__device__ __forceinline__ void chunks_body(const SpmmArgs &p, const lgc_chunk *__restrict__ chunks, int32_t n_chunks,
                                            float *__restrict__ partials, int64_t block) {
    const int64_t wave = block * (kBlock / kWave) + (threadIdx.x / kWave);
    if (wave >= n_chunks) return;  // wave-uniform
    chunk_core<VEC, FILTER>(p, chunks[wave], partials);
}

// The chunk part of a seeded pull with row marks (lgc_seed_pull, row_mark != NULL): a wavefront screens kScreen chunks at
// once -- lane l reads chunk l's descriptor and its row's mark --, writes the zero partial (or zero row) of each unmarked
// one as a whole line and then works through the marked ones, all lanes on one chunk at a time.  One wavefront per chunk spent three
// dependent loads on each of 60 k chunks to find the 11 % worth reading (100 us per training step).
constexpr int kScreen = 16;
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_seed_pull_chunks(SpmmArgs p, const lgc_chunk *__restrict__ chunks, int32_t n_chunks,
                                                            float *__restrict__ partials) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x / kWave);
    // lane l looks at chunk l * n_waves + wave: the chunks of one long (hence surely marked) row go to different wavefronts
    const int64_t n_waves = ((int64_t)n_chunks + kScreen - 1) / kScreen;
    const int64_t c = (int64_t)lane * n_waves + wave;
    const bool valid = lane < kScreen && wave < n_waves && c < n_chunks;
    lgc_chunk ch = {0, 0, 0, -1};
    bool marked = false;
    if (valid) {
        ch = chunks[c];
        marked = p.row_mark[ch.row] != 0;
    }
    unsigned long long zero = __ballot(valid && !marked);   // no seed among the row's columns: zeros, unread --
    while (zero != 0) {                                      // one whole-line store per chunk by all lanes
        const int j = __builtin_ctzll(zero);
        zero &= zero - 1;
        const int row = __shfl(ch.row, j), slot = __shfl(ch.slot, j);
        float *dst = slot >= 0 ? partials + (int64_t)slot * p.dim : p.y + (int64_t)row * p.y_stride;
        for (int i = lane; i < p.dim; i += kWave) dst[i] = 0.0f;
    }
    unsigned long long todo = __ballot(marked);
    while (todo != 0) {   // wave-uniform
        const int j = __builtin_ctzll(todo);
        todo &= todo - 1;
        const lgc_chunk cj = {__shfl(ch.row, j), __shfl(ch.begin, j), __shfl(ch.end, j), __shfl(ch.slot, j)};
        chunk_core<VEC, true, true>(p, cj, partials);
    }
}

// A handful of rows given by id (lgc_spmm_rows): one wavefront per listed row.  A training step scores 2B label pairs, so
// of the last user step's 1.6 M output rows only the <= 2B rows of the batch's users are ever read (src/lightgcn.py:123-125).
// Rows of up to 32 entries are summed by lane group 0 in entry order -- the same association, hence the same bits, as the
// tiled kernels and the row-pointer kernel; longer rows are strided over the lane groups and reduced in group order.
template <int VEC>
__device__ __forceinline__ void listed_row_body(const SpmmArgs &p, const int64_t row) {
    const int lane = threadIdx.x & (kWave - 1);
    const int groups = kWave / p.lpr;
    const int g = lane / p.lpr;
    const int l = lane - g * p.lpr;
    const int c0 = lane_column<VEC>(l, p.dim);
    const bool active = g < groups;
    const int32_t s = p.rowptr[row], e = p.rowptr[row + 1];
    const bool short_row = e - s <= 32;                   // wave-uniform
    const int stride = short_row ? 1 : groups;
    Acc<VEC> acc;
    acc.zero();
    if (active && (!short_row || g == 0)) {
        const float *xb = p.x + c0;
        const lgc_entry *__restrict__ ent = p.entries;
        int32_t k = s + (short_row ? 0 : g);
        for (; k + 3 * stride < e; k += 4 * stride) {
            const lgc_entry e0 = ent[k], e1 = ent[k + stride], e2 = ent[k + 2 * stride], e3 = ent[k + 3 * stride];
            Acc<VEC> x0 = load_row<VEC>(xb + (int64_t)e0.col * p.x_stride);
            Acc<VEC> x1 = load_row<VEC>(xb + (int64_t)e1.col * p.x_stride);
            Acc<VEC> x2 = load_row<VEC>(xb + (int64_t)e2.col * p.x_stride);
            Acc<VEC> x3 = load_row<VEC>(xb + (int64_t)e3.col * p.x_stride);
            mul_add<VEC>(acc, e0.val, x0);
            mul_add<VEC>(acc, e1.val, x1);
            mul_add<VEC>(acc, e2.val, x2);
            mul_add<VEC>(acc, e3.val, x3);
        }
        for (; k < e; k += stride) {
            const lgc_entry e0 = ent[k];
            Acc<VEC> x0 = load_row<VEC>(xb + (int64_t)e0.col * p.x_stride);
            mul_add<VEC>(acc, e0.val, x0);
        }
    }
    if (!short_row) {     // every lane executes the shuffles; only group 0 keeps the result
        for (int j = 1; j < groups; ++j) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float other = __shfl_down(acc.v[i], j * p.lpr);
                if (g == 0) acc.v[i] = __fadd_rn(acc.v[i], other);
            }
        }
    }
    if (!active || g != 0) return;
    Acc<VEC> rv;
    rv.zero();
    if (p.r != nullptr) rv = load_row<VEC>(p.r + row * p.r_stride + c0);
    finish_row<VEC, SpmmArgs>(p, row, c0, acc, rv);
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_spmm_rows(SpmmArgs p, const int64_t *__restrict__ row_ids, int64_t n_ids) {
    const int64_t m = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x / kWave);
    if (m >= n_ids) return;  // wave-uniform
    const int64_t row = row_ids[m];
    if (row < p.row_begin || row >= p.row_end) return;   // not a row of this operator half: skipped
    listed_row_body<VEC>(p, row);
}

// Listed rows of ANY length (lgc_spmm_rows_split).  The rows a training step scores on the item side are item rows: 186
// entries on average, the hubs beyond 10^5 -- one wavefront per row would leave the step waiting for the wavefront that
// walks a hub.  The list is therefore cut into chunks ON THE DEVICE (no host round trip, fixed launch shapes, so the
// step can be recorded as a HIP graph): k_rows_plan (ONE workgroup) counts the entries of the listed rows, picks the
// chunk length -- kRowsChunk entries, or longer if the caller's partial table would not hold that many chunks -- and writes every
// list position's first chunk number (an exclusive scan; a row of up to 32 entries, or an empty one, is one chunk; a
// position whose id lies outside the operator half has none); k_rows_chunks: a fixed grid of wavefronts strides over the
// chunk numbers, finds the list position that owns a chunk (a 64-ary search: the lanes probe 64 offsets at once) and
// either finishes the row (one chunk: rows of up to 32 entries in entry order, the bits of the tiled kernels) or leaves
// a partial row; k_rows_combine adds the partial rows of a position in chunk order and applies the epilogue.  A repeated
// id is computed once per occurrence and written with the same bits.
constexpr int kRowsPlanBlock = 1024;
#ifndef LGC_ROWS_CHUNK            // (-DLGC_ROWS_CHUNK=128 builds the A/B variant of profiles/r04g)
#define LGC_ROWS_CHUNK 256
#endif
constexpr int kRowsChunk = LGC_ROWS_CHUNK;   // 128: the same step time (five interleaved pairs)

__global__ __launch_bounds__(kRowsPlanBlock) void k_rows_plan(const int32_t *__restrict__ rowptr, const int64_t *__restrict__ row_ids,
                                                              int64_t n_ids, int32_t row_begin, int32_t row_end, int64_t cap,
                                                              int32_t *__restrict__ work) {
    constexpr int kWaves = kRowsPlanBlock / kWave;
    __shared__ long long s_sum[kWaves];
    __shared__ int s_scan[kWaves];
    __shared__ int s_len, s_tile;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    long long sum = 0;
    for (int64_t m = tid; m < n_ids; m += kRowsPlanBlock) {
        const int64_t row = row_ids[m];
        if (row >= row_begin && row < row_end) sum += rowptr[row + 1] - rowptr[row];
    }
    for (int off = kWave / 2; off > 0; off >>= 1) sum += __shfl_down(sum, off);
    if (lane == 0) s_sum[wv] = sum;
    __syncthreads();
    if (tid == 0) {
        long long total = 0;
        for (int i = 0; i < kWaves; ++i) total += s_sum[i];
        // sum_i ceil(len_i / L) <= total / L + n_ids  must fit the partial table: L >= total / (cap - n_ids)
        const long long room = cap - n_ids;          // >= 1 (checked by the host)
        long long len = kRowsChunk;
        if (total / kRowsChunk > room) len = ((total + room - 1) / room + kWave - 1) / kWave * kWave;
        s_len = (int)(len > 0x40000000 ? 0x40000000 : len);
    }
    __syncthreads();
    const int chunk_len = s_len;
    int base = 0;
    for (int64_t m0 = 0; m0 < n_ids; m0 += kRowsPlanBlock) {
        const int64_t m = m0 + tid;
        int c = 0;
        if (m < n_ids) {
            const int64_t row = row_ids[m];
            if (row >= row_begin && row < row_end) {
                const int len = rowptr[row + 1] - rowptr[row];
                c = len <= 32 ? 1 : (len + chunk_len - 1) / chunk_len;
            }
        }
        int incl = c;
        for (int off = 1; off < kWave; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == kWave - 1) s_scan[wv] = incl;
        __syncthreads();
        if (wv == 0) {
            const int v = lane < kWaves ? s_scan[lane] : 0;
            int iv = v;
            for (int off = 1; off < kWaves; off <<= 1) {
                const int t = __shfl_up(iv, off);
                if (lane >= off) iv += t;
            }
            if (lane < kWaves) s_scan[lane] = iv - v;
            if (lane == kWaves - 1) s_tile = iv;
        }
        __syncthreads();
        if (m < n_ids) work[m] = base + s_scan[wv] + incl - c;
        base += s_tile;
        __syncthreads();
    }
    if (tid == 0) {
        work[n_ids] = base;
        work[n_ids + 1] = chunk_len;
    }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_rows_chunks(SpmmArgs p, const int64_t *__restrict__ row_ids, int64_t n_ids,
                                                       const int32_t *__restrict__ work, float *__restrict__ partials,
                                                       int32_t compact) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t n_waves = (int64_t)gridDim.x * (kBlock / kWave);
    const int32_t total = work[n_ids], chunk_len = work[n_ids + 1];
    for (int64_t w = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x / kWave); w < total; w += n_waves) {   // wave-uniform
        // the last list position m with work[m] <= w (then w < work[m + 1]: that position owns chunk w); invariant
        // work[lo] <= w < work[hi], the lanes probe the range at 64 evenly spaced offsets per round
        int64_t lo = 0, hi = n_ids;
        while (hi - lo > 1) {
            const int64_t step = (hi - lo + kWave - 1) / kWave;
            const int64_t at = lo + (int64_t)lane * step;
            const bool le = at < hi && work[at] <= (int32_t)w;
            const int k = __popcll(__ballot(le)) - 1;            // lane 0 probes lo itself: at least one bit
            lo += (int64_t)k * step;
            hi = min(lo + step, hi);
        }
        const int64_t m = lo;
        const int64_t row = row_ids[m];
        const int32_t first = work[m], count = work[m + 1] - first;
        const int32_t s = p.rowptr[row], e = p.rowptr[row + 1];
        SpmmArgs q = p;
        if (compact) q.y = p.y + (m - row) * p.y_stride;          // y[row] of q is y[m] of p
        if (count == 1) {
            if (e - s <= 32) {
                listed_row_body<VEC>(q, row);
            } else {
                const lgc_chunk ch = {(int32_t)row, s, e, -1};
                chunk_core<VEC>(q, ch, partials);
            }
        } else {
            const int32_t begin = s + (int32_t)(w - first) * chunk_len;
            const lgc_chunk ch = {(int32_t)row, begin, min(begin + chunk_len, e), (int32_t)w};
            chunk_core<VEC>(q, ch, partials);
        }
    }
}

// One launch per operator and hop: the first `chunk_blocks` workgroups take the chunk work list (long rows
// are dispatched first, they run longest), the rest the short rows.  Launching the two parts separately
// left each one's ramp-up and tail exposed -- 7 back-to-back launches per hop of 4-45 us each on a rank of
// an 8-way partition.  The row part here reads its entries through the row pointer; it serves widths below 4 and
// callers that ask for it -- rows of up to 32 entries normally go through the tiled kernels (lgc_spmm_tiles).
template <int VEC, bool FILTER = false>
__global__ __launch_bounds__(kBlock) void k_spmm_hop(SpmmArgs p, const lgc_chunk *__restrict__ chunks,
                                                    int32_t n_chunks, float *__restrict__ partials,
                                                    int32_t chunk_blocks) {
    if ((int32_t)blockIdx.x < chunk_blocks) {
        chunks_body<VEC, FILTER>(p, chunks, n_chunks, partials, blockIdx.x);
    } else {
        const int64_t block = (int64_t)blockIdx.x - chunk_blocks;
        rows_body<VEC, FILTER>(p, block);
    }
}

// Rows cut into several chunks: one wavefront per row.  Lane group g adds slots g, g+G, g+2G, ...
// (4 loads in flight per group), then the G group sums are added in group order -- a fixed
// association, so the result does not depend on scheduling.
template <int VEC>
__device__ __forceinline__ void combine_core(const SpmmArgs &p, const lgc_multi_row mr, const float *__restrict__ partials) {
    const int lane = threadIdx.x & (kWave - 1);
    const int groups = kWave / p.lpr;
    const int g = lane / p.lpr;
    const int l = lane - g * p.lpr;
    const int c0 = lane_column<VEC>(l, p.dim);
    const bool active = g < groups;
    Acc<VEC> acc;
    acc.zero();
    if (active) {
        const float *pb = partials + c0;
        int32_t s = mr.slot_begin + g;
        for (; s + 3 * groups < mr.slot_end; s += 4 * groups) {
            Acc<VEC> t0 = load_row<VEC>(pb + (int64_t)s * p.dim);
            Acc<VEC> t1 = load_row<VEC>(pb + (int64_t)(s + groups) * p.dim);
            Acc<VEC> t2 = load_row<VEC>(pb + (int64_t)(s + 2 * groups) * p.dim);
            Acc<VEC> t3 = load_row<VEC>(pb + (int64_t)(s + 3 * groups) * p.dim);
#pragma unroll
            for (int i = 0; i < VEC; ++i)
                acc.v[i] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc.v[i], t0.v[i]), t1.v[i]), t2.v[i]), t3.v[i]);
        }
        for (; s < mr.slot_end; s += groups) {
            Acc<VEC> t0 = load_row<VEC>(pb + (int64_t)s * p.dim);
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc.v[i] = __fadd_rn(acc.v[i], t0.v[i]);
        }
    }
    for (int j = 1; j < groups; ++j) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
            float other = __shfl_down(acc.v[i], j * p.lpr);
            if (g == 0) acc.v[i] = __fadd_rn(acc.v[i], other);
        }
    }
    if (active && g == 0) {
        Acc<VEC> rv;
        rv.zero();
        if (p.r != nullptr) rv = load_row<VEC>(p.r + (int64_t)mr.row * p.r_stride + c0);
        finish_row<VEC, SpmmArgs>(p, mr.row, c0, acc, rv);
    }
}

template <int VEC>
__global__ __launch_bounds__(kBlock) void k_spmm_combine(SpmmArgs p, const lgc_multi_row *__restrict__ multi,
                                                        int32_t n_multi, const float *__restrict__ partials) {
    const int64_t m = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x / kWave);
    if (m >= n_multi) return;  // wave-uniform
    combine_core<VEC>(p, multi[m], partials);
}

// lgc_spmm_rows_split, last launch: one wavefront per list position that was cut into several chunks.
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_rows_combine(SpmmArgs p, const int64_t *__restrict__ row_ids, int64_t n_ids,
                                                        const int32_t *__restrict__ work, const float *__restrict__ partials,
                                                        int32_t compact) {
    const int64_t m = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x / kWave);
    if (m >= n_ids) return;  // wave-uniform
    const int32_t first = work[m], last = work[m + 1];
    if (last - first <= 1) return;   // no chunk (foreign id) or finished by k_rows_chunks
    const int64_t row = row_ids[m];
    SpmmArgs q = p;
    if (compact) q.y = p.y + (m - row) * p.y_stride;
    const lgc_multi_row mr = {(int32_t)row, first, last};
    combine_core<VEC>(q, mr, partials);
}


// ----------------------------------------------------------------------------------------
// Rows with at most 32 entries, tiled: k_rows_tile
// ----------------------------------------------------------------------------------------
// A wave lives for several TILES.  A tile is 1 KiB * L of row heads laid out so that ONE coalesced dwordx4
// per lane (per load k < L) brings the entries of R = 128 L / W rows, plus R row ids (the rows are listed in
// a processing order chosen for L2 locality, so ids are explicit).  The next tile's loads are issued before the
// current tile's gathers: a wave pays the HBM latency of the streamed operands once, not once per 4 rows --
// the old one-batch-per-wave kernel kept a wave slot busy ~6 us for 1.3 KB of traffic whatever the gathers hit
// (tools/exp_floor.py).  No row pointer is read: padding entries carry col = -1.
//
struct TileArgs {
    const int32_t *order;   // [n_tiles * R] row ids in processing order, -1 = padding slot
    const int32_t *meta;    // [n_tiles] 4 x 8 bits: longest row of batch 0..3 of the tile (fast path only)
    const u4 *slab;         // [n_tiles * L * 64] 16-byte pieces = 2 entries each
    uint32_t x_bytes, y_bytes, r_bytes;   // table sizes for the buffer descriptors of the fast path
    const float *x;
    float *y;
    const float *r;
    int64_t x_stride, y_stride, r_stride;
    float a, b;
    int32_t n_tiles, tiles_per_wave;
    int32_t dim, lpr;       // columns, lanes per row
    int32_t wt_store;
};

template <int W, int L>
__device__ __forceinline__ void tiles_body(const TileArgs &p, const int64_t block) {
    constexpr int R = 128 * L / W;     // rows per tile
    constexpr int Wk = W / L;          // entries of a row in one load
    constexpr int PPR = Wk / 2;        // 16-byte pieces of a row in one load
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = block * (kBlock / kWave) + (threadIdx.x / kWave);
    int64_t tile = wave * p.tiles_per_wave;
    if (tile >= p.n_tiles) return;  // wave-uniform
    const int64_t tile_end = min(tile + (int64_t)p.tiles_per_wave, (int64_t)p.n_tiles);
    const int dim = p.dim;
    const int G = kWave / p.lpr;       // rows per batch
    const int g = lane / p.lpr;
    const int l = lane - g * p.lpr;
    const bool lane_on = g < G;
    const int c0 = min(l * 4, dim - 4);
    const float *xb = p.x + c0;

    u4 nxt[L];
    int32_t nxt_rid = -1;
#pragma unroll
    for (int k = 0; k < L; ++k) nxt[k] = p.slab[(tile * L + k) * kWave + lane];
    if (lane < R) nxt_rid = p.order[tile * R + lane];
    for (; tile < tile_end; ++tile) {
        u4 cur[L];
#pragma unroll
        for (int k = 0; k < L; ++k) cur[k] = nxt[k];
        const int32_t cur_rid = nxt_rid;
        if (tile + 1 < tile_end) {
#pragma unroll
            for (int k = 0; k < L; ++k) nxt[k] = p.slab[((tile + 1) * L + k) * kWave + lane];
            if (lane < R) nxt_rid = p.order[(tile + 1) * R + lane];
        }
        for (int s0 = 0; s0 < R; s0 += G) {
            const int s = s0 + g;
            const bool slot_on = lane_on && s < R;
            const int ss = slot_on ? s : 0;
            const int32_t row = __shfl(cur_rid, ss);
            const bool on = slot_on && row >= 0;
            Acc<4> acc, rv;
            acc.zero();
            rv.zero();
            if (on && p.r != nullptr) rv = load_row<4>(p.r + (int64_t)row * p.r_stride + c0);
            bool more = true;
#pragma unroll
            for (int k = 0; k < L; ++k) {
#pragma unroll
                for (int j0 = 0; j0 < Wk; j0 += 4) {
                    if (!more) continue;
                    int32_t col[4];
                    float val[4];
                    Acc<4> xv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {  // every lane of the wave executes the shuffles
                        const int src = ss * PPR + ((j0 + j) >> 1);
                        col[j] = __shfl((int)(((j0 + j) & 1) ? cur[k].z : cur[k].x), src);
                        val[j] = __int_as_float(__shfl((int)(((j0 + j) & 1) ? cur[k].w : cur[k].y), src));
                    }
                    // entries are packed at the front of a row: when no row of the batch has entry j0, none has more
                    more = __ballot(on && col[0] >= 0) != 0;
                    if (!more) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (on && col[j] >= 0) xv[j] = load_row<4>(xb + (int64_t)col[j] * p.x_stride);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (on && col[j] >= 0) mul_add<4>(acc, val[j], xv[j]);
                }
            }
            if (on) finish_row<4, TileArgs>(p, row, c0, acc, rv);
        }
    }
}

template <int W, int L>
__global__ __launch_bounds__(kBlock) void k_rows_tile(TileArgs p) {
    tiles_body<W, L>(p, blockIdx.x);
}


// Fast path for rows of 13..16 lanes (D = 49..64): no divergent control flow and ~7 VALU instructions per
// gathered row instead of ~20 -- the bpermute version is bound by vector-ALU issue, not by memory
// (profiles/r02_*: SQ busy 90 %, 20 VALU per VMEM instruction).
//   * a lane group is one DPP row of 16 lanes and owns the pieces of ITS rows (slot 4g + bt = pieces
//     16g + 4bt ..), so an entry reaches the group's lanes with v_mov_b32 row_newbcast:k, no LDS crossbar;
//   * gathers and stores go through buffer descriptors with 32-bit offsets = mad24(row, stride, lane offset);
//     a padding entry / padding row has the 24-bit id 0xFFFFFF, whose offset the host has checked to lie
//     beyond the table: the hardware returns zeros for the load and drops the store;
//   * how many entries a batch needs is wave-uniform metadata -> scalar branches.
// Cache policy of the tiled rows' output stores: nt (2).  Alone, nt stores are the slowest flavour (5.6 TB/s against
// 8.5-8.7 for plain / sc1, tools/microbench/line_cost.hip), but beside gathers they cost the least: the user step takes
// 240 us with nt, 245 with sc1 / sc0 sc1, 249 plain (plain also lets the written lines evict gathered rows: 1.14 GB
// fetched instead of 1.00, profiles/r03a_variants_traffic.txt).
constexpr int kTileStoreAux = 2;

template <int K>
__device__ __forceinline__ int bcast16(int v) {
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xf, 0xf, true);   // row_newbcast:K (gfx90a+)
}

typedef float f2 __attribute__((ext_vector_type(2)));

// (col, vb) = (colbits, valbits) of lane IDX of every DPP row; IDX is a compile-time constant after unrolling
#define LGC_BCAST16(IDX, colbits, valbits, col, vb)                                                                    \
    switch (IDX) {                                                                                                     \
        case 0: col = bcast16<0>(colbits); vb = bcast16<0>(valbits); break;                                            \
        case 1: col = bcast16<1>(colbits); vb = bcast16<1>(valbits); break;                                            \
        case 2: col = bcast16<2>(colbits); vb = bcast16<2>(valbits); break;                                            \
        case 3: col = bcast16<3>(colbits); vb = bcast16<3>(valbits); break;                                            \
        case 4: col = bcast16<4>(colbits); vb = bcast16<4>(valbits); break;                                            \
        case 5: col = bcast16<5>(colbits); vb = bcast16<5>(valbits); break;                                            \
        case 6: col = bcast16<6>(colbits); vb = bcast16<6>(valbits); break;                                            \
        case 7: col = bcast16<7>(colbits); vb = bcast16<7>(valbits); break;                                            \
        case 8: col = bcast16<8>(colbits); vb = bcast16<8>(valbits); break;                                            \
        case 9: col = bcast16<9>(colbits); vb = bcast16<9>(valbits); break;                                            \
        case 10: col = bcast16<10>(colbits); vb = bcast16<10>(valbits); break;                                         \
        case 11: col = bcast16<11>(colbits); vb = bcast16<11>(valbits); break;                                         \
        case 12: col = bcast16<12>(colbits); vb = bcast16<12>(valbits); break;                                         \
        case 13: col = bcast16<13>(colbits); vb = bcast16<13>(valbits); break;                                         \
        case 14: col = bcast16<14>(colbits); vb = bcast16<14>(valbits); break;                                         \
        default: col = bcast16<15>(colbits); vb = bcast16<15>(valbits); break;                                         \
    }


template <int W, int L>
__device__ __forceinline__ void tiles_body_dpp(const TileArgs &p, const int64_t block) {
    constexpr int R = 128 * L / W;     // rows per tile
    constexpr int Wk = W / L;          // entries of a row in one load
    constexpr int PPR = Wk / 2;        // 16-byte pieces of a row in one load
    constexpr int B = R / 4;           // batches of 4 rows
    static_assert(B * PPR == 16, "a lane group's rows fill its 16 lanes");
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = block * (kBlock / kWave) + (threadIdx.x / kWave);
    int64_t tile = __builtin_amdgcn_readfirstlane((int)(wave * p.tiles_per_wave));
    if (tile >= p.n_tiles) return;  // wave-uniform
    const int64_t tile_end = min(tile + (int64_t)p.tiles_per_wave, (int64_t)p.n_tiles);
    const int l = lane & 15;
    const int c0 = min(l * 4, p.dim - 4);
    const auto xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, p.x_bytes, 0x00020000);
    const auto ysrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, p.y_bytes, 0x00020000);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.r, 0, p.r != nullptr ? p.r_bytes : 0u, 0x00020000);
    const unsigned xoff = (unsigned)c0 * 4u;    // lpr == 16: every lane of the group owns a float4 of the row
    const unsigned xs = (unsigned)p.x_stride * 4u, ys = (unsigned)p.y_stride * 4u, rs = (unsigned)p.r_stride * 4u;

    u4 nxt[L];
    int32_t nxt_rid = -1;
#pragma unroll
    for (int k = 0; k < L; ++k) nxt[k] = __builtin_nontemporal_load(p.slab + (tile * L + k) * kWave + lane);
    if (l < B) nxt_rid = p.order[tile * R + (lane >> 4) * B + l];
    int32_t nxt_meta = p.meta[tile];
    for (; tile < tile_end; ++tile) {
        u4 cur[L];
#pragma unroll
        for (int k = 0; k < L; ++k) cur[k] = nxt[k];
        const int32_t cur_rid = nxt_rid;
        const int32_t meta = nxt_meta;
        if (tile + 1 < tile_end) {
#pragma unroll
            for (int k = 0; k < L; ++k) nxt[k] = __builtin_nontemporal_load(p.slab + ((tile + 1) * L + k) * kWave + lane);
            if (l < B) nxt_rid = p.order[(tile + 1) * R + (lane >> 4) * B + l];
            nxt_meta = p.meta[tile + 1];
        }
#pragma unroll
        for (int bt = 0; bt < B; ++bt) {
            const int nmax = (meta >> (8 * bt)) & 0xFF;   // scalar
            const int row24 = bt == 0 ? bcast16<0>(cur_rid) : bt == 1 ? bcast16<1>(cur_rid)
                              : bt == 2 ? bcast16<2>(cur_rid) : bcast16<3>(cur_rid);
            f2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
            f4 rv = {0.0f, 0.0f, 0.0f, 0.0f};
            if (p.r != nullptr)
                rv = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, __umul24(row24, rs) + xoff, 0, 0));
#pragma unroll
            for (int k = 0; k < L; ++k) {
#pragma unroll
                for (int j0 = 0; j0 < Wk; j0 += 4) {
                    if (k * Wk + j0 < nmax) {     // wave-uniform
                        f4 xv[4];
                        float val[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int colbits = ((j0 + j) & 1) ? (int)cur[k].z : (int)cur[k].x;
                            const int valbits = ((j0 + j) & 1) ? (int)cur[k].w : (int)cur[k].y;
                            int col, vb;
                            LGC_BCAST16(bt * PPR + ((j0 + j) >> 1), colbits, valbits, col, vb)   // compile-time lane
                            val[j] = __int_as_float(vb);
                            xv[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, __umul24(col, xs) + xoff, 0, 0));
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) {   // product rounded, then added (no FMA: -ffp-contract=off)
                            const f2 v2 = {val[j], val[j]};
                            const f2 lo = {xv[j].x, xv[j].y}, hi = {xv[j].z, xv[j].w};
                            a0 = a0 + lo * v2;
                            a1 = a1 + hi * v2;
                        }
                    }
                }
            }
            Acc<4> acc;
            acc.v[0] = a0.x; acc.v[1] = a0.y; acc.v[2] = a1.x; acc.v[3] = a1.y;
            if (p.a != 1.0f) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc.v[i] = __fmul_rn(p.a, acc.v[i]);
            }
            if (p.r != nullptr) {
                acc.v[0] = __fadd_rn(acc.v[0], __fmul_rn(p.b, rv.x));
                acc.v[1] = __fadd_rn(acc.v[1], __fmul_rn(p.b, rv.y));
                acc.v[2] = __fadd_rn(acc.v[2], __fmul_rn(p.b, rv.z));
                acc.v[3] = __fadd_rn(acc.v[3], __fmul_rn(p.b, rv.w));
            }
            const f4 o = {acc.v[0], acc.v[1], acc.v[2], acc.v[3]};
            // the output row is not read again in this launch (the host made sure of < 4 GiB): cache policy kTileStoreAux.
            // Holding a tile's four output rows back until its last batch, or keeping a batch's eight gathers in flight
            // together, changes nothing (539.7 / 539.6 / 539.5 us per hop, profiles/r03h_defer_stores_batch8.txt)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o), ysrc, __umul24(row24, ys) + xoff, 0, kTileStoreAux);
        }
    }
}

// The same for rows of 68..128 columns (D = 80, 90, 96, 128): a row takes TWO DPP rows of a wavefront -- lanes
// 0-15 its columns 0..63, lanes 16-31 the rest -- so a wave carries two rows per instruction and a 16-row tile takes
// two passes.  Both DPP rows of a pair load the same 16 pieces of the tile, so the row_newbcast hand-out is
// unchanged; lanes beyond the row's width use the out-of-range id for every load and store.
template <int W, int L>
__device__ __forceinline__ void tiles_body_dpp_wide(const TileArgs &p, const int64_t block) {
    constexpr int R = 128 * L / W;     // rows per tile
    constexpr int Wk = W / L;          // entries of a row in one load
    constexpr int PPR = Wk / 2;        // 16-byte pieces of a row in one load
    constexpr int B = R / 4;           // row slots per 16 pieces
    static_assert(B * PPR == 16, "four row slots' pieces fill a DPP row");
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t wave = block * (kBlock / kWave) + (threadIdx.x / kWave);
    int64_t tile = __builtin_amdgcn_readfirstlane((int)(wave * p.tiles_per_wave));
    if (tile >= p.n_tiles) return;  // wave-uniform
    const int64_t tile_end = min(tile + (int64_t)p.tiles_per_wave, (int64_t)p.n_tiles);
    const int l = lane & 15, pair = lane >> 5, half = (lane >> 4) & 1;
    const int dim = p.dim;
    const int rest = dim - 64;                                   // 4 .. 64 columns in the second DPP row
    const bool lane_on = half == 0 || l * 4 < rest;              // ceil(rest / 4) lanes, the last one overlapping
    const int c0 = half == 0 ? l * 4 : 64 + min(l * 4, rest - 4);
    const unsigned xoff = (unsigned)c0 * 4u;
    const int pad24 = 0xFFFFFF;
    const auto xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, p.x_bytes, 0x00020000);
    const auto ysrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.y, 0, p.y_bytes, 0x00020000);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.r, 0, p.r != nullptr ? p.r_bytes : 0u, 0x00020000);
    const unsigned xs = (unsigned)p.x_stride * 4u, ys = (unsigned)p.y_stride * 4u, rs = (unsigned)p.r_stride * 4u;
    for (; tile < tile_end; ++tile) {
        const int32_t meta = p.meta[tile];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int quad = 2 * pass + pair;                    // which 16 pieces / which B row slots of the tile
            u4 cur[L];
#pragma unroll
            for (int k = 0; k < L; ++k) cur[k] = __builtin_nontemporal_load(p.slab + (tile * L + k) * kWave + quad * 16 + l);
            int32_t cur_rid = -1;
            if (l < B) cur_rid = p.order[tile * R + quad * B + l];
#pragma unroll
            for (int bt = 0; bt < B; ++bt) {
                const int nmax = (meta >> (8 * bt)) & 0xFF;   // scalar (an upper bound: the longest of four slots)
                int row24 = bt == 0 ? bcast16<0>(cur_rid) : bt == 1 ? bcast16<1>(cur_rid)
                            : bt == 2 ? bcast16<2>(cur_rid) : bcast16<3>(cur_rid);
                row24 = lane_on ? row24 : pad24;
                f2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
                f4 rv = {0.0f, 0.0f, 0.0f, 0.0f};
                if (p.r != nullptr)
                    rv = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, __umul24(row24, rs) + xoff, 0, 0));
#pragma unroll
                for (int k = 0; k < L; ++k) {
#pragma unroll
                    for (int j0 = 0; j0 < Wk; j0 += 4) {
                        if (k * Wk + j0 < nmax) {     // wave-uniform
                            f4 xv[4];
                            float val[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int colbits = ((j0 + j) & 1) ? (int)cur[k].z : (int)cur[k].x;
                                const int valbits = ((j0 + j) & 1) ? (int)cur[k].w : (int)cur[k].y;
                                int col, vb;
                                LGC_BCAST16(bt * PPR + ((j0 + j) >> 1), colbits, valbits, col, vb)
                                col = lane_on ? col : pad24;
                                val[j] = __int_as_float(vb);
                                xv[j] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(xsrc, __umul24(col, xs) + xoff, 0, 0));
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) {   // product rounded, then added (no FMA: -ffp-contract=off)
                                const f2 v2 = {val[j], val[j]};
                                const f2 lo = {xv[j].x, xv[j].y}, hi = {xv[j].z, xv[j].w};
                                a0 = a0 + lo * v2;
                                a1 = a1 + hi * v2;
                            }
                        }
                    }
                }
                Acc<4> acc;
                acc.v[0] = a0.x; acc.v[1] = a0.y; acc.v[2] = a1.x; acc.v[3] = a1.y;
                if (p.a != 1.0f) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc.v[i] = __fmul_rn(p.a, acc.v[i]);
                }
                if (p.r != nullptr) {
                    acc.v[0] = __fadd_rn(acc.v[0], __fmul_rn(p.b, rv.x));
                    acc.v[1] = __fadd_rn(acc.v[1], __fmul_rn(p.b, rv.y));
                    acc.v[2] = __fadd_rn(acc.v[2], __fmul_rn(p.b, rv.z));
                    acc.v[3] = __fadd_rn(acc.v[3], __fmul_rn(p.b, rv.w));
                }
                const f4 o = {acc.v[0], acc.v[1], acc.v[2], acc.v[3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, o), ysrc, __umul24(row24, ys) + xoff, 0, kTileStoreAux);
            }
        }
    }
}

template <int W, int L>
__global__ __launch_bounds__(kBlock) void k_rows_tile_dpp(TileArgs p) {
    if (p.dim > 64) tiles_body_dpp_wide<W, L>(p, blockIdx.x);
    else tiles_body_dpp<W, L>(p, blockIdx.x);
}


// ----------------------------------------------------------------------------------------
// Long rows over a huge gathered table: the band sweep (k_sweep) -- item step of a user|item graph
// ----------------------------------------------------------------------------------------
// The item step gathers 10.2 M user rows from a 420 MB table; with one wavefront per chunk of one row every
// use of a user row (6.2 on average, by unrelated rows at unrelated times on unrelated XCDs) crosses the fabric:
// 2.75 GB per hop at the fabric's ~7 TB/s, although the same kernel runs in 180 us when the gathers hit L2
// (tools/exp_floor.py).  Here the columns are cut into 8 bands = XCDs (blockIdx % 8), every (row, band) pair is a
// PIECE with an accumulator in LDS, and all wavefronts of a band walk its columns in ascending order at the same
// pace, so a table row is fetched from memory by the first wave that needs it and found in that XCD's L2 by the
// others.  LDS holds 16 x 39 accumulators per CU, a band has ~56 k pieces: three rounds.
//   * a wave owns up to `row_cap` pieces and one merged, column-sorted entry list over all of them, cut into STEPS
//     of four entries (one per 16-lane group) that the planner keeps free of two entries for the same piece, so the
//     four read-modify-writes of a step never touch the same LDS row; entry = {col : 24 | piece : 8, val};
//   * 32 steps = one 1 KiB slab, fetched by one coalesced dwordx4 per lane; lane 16 g + q holds steps 2q, 2q+1 of
//     group g, handed to the group's lanes with DPP row_newbcast; gathers go through a buffer descriptor, padding
//     entries (col = 0xFFFFFF, out of range: zeros) update a dummy LDS row;
//   * at the end a wave writes its pieces to their partial slots (contiguous per output row, band-major) and
//     k_spmm_combine adds a row's slots in that fixed order: deterministic, no float atomics.
// Gathers a wavefront of the row sweep keeps in flight: 8 slots, of which the compiler fills seven under the 64-register
// budget.  Both directions lose: four in flight 48 % L2 hits and 247 us, a full eight 41 % and 249-263 us, sixteen 301 us,
// against 52-53 % and 241-245 us (profiles/r03a, r03j, r03k) -- how far the 256 wavefronts of a band drift apart, and with
// it the L2 hit rate, depends on it more than the memory-level parallelism does.
constexpr int kSweepDepth = 8;

struct SweepArgs {
    const u4 *slabs;               // [n_slabs * 64]
    const int32_t *wave_slab_ptr;  // [n_waves + 1]
    const int32_t *wave_npieces;   // [n_waves]
    const int32_t *piece_slot;     // [n_waves * row_cap]
    const float *x;
    float *partials;               // [n_slots, dim]
    int64_t x_stride;
    uint32_t x_bytes;
    int32_t dim, row_cap, n_waves, wave_begin;
    unsigned long long *trace;     // diagnostics only (LGCN_SWEEP_TRACE): [n_waves, 16] s_memrealtime stamps per slab
    int32_t pstride, pcol;         // partial rows are pstride floats, this launch fills columns pcol .. pcol + dim
};

// A wavefront's piece -> partial-slot table, read ONCE at its start (lane i keeps pieces i, 64 + i, ...) so that the
// write-out at its end issues its stores back to back: fetched inside that loop, every iteration waited for a dependent
// load -- and, vmcnt retiring in order, for the previous iteration's store to HBM.
struct PieceSlots {
    int s[4];                                   // row_cap <= 254
    __device__ __forceinline__ void load(const int32_t *slots, int npieces, int lane) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s[j] = (j * 64 + lane < npieces) ? slots[j * 64 + lane] : 0;
    }
    __device__ __forceinline__ int at(int pc) const {   // pc may differ between lanes
        int v = __shfl(s[0], pc & 63);
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            const int o = __shfl(s[j], pc & 63);
            v = (pc >> 6) == j ? o : v;
        }
        return v;
    }
};

template <int Q, int HALF>
__device__ __forceinline__ void sweep_fetch(const u4 &cur, int &packed, int &valbits) {
    packed = bcast16<Q>((int)(HALF ? cur.z : cur.x));
    valbits = bcast16<Q>((int)(HALF ? cur.w : cur.y));
}

template <int DEPTH>
// The register budget stays at 64 although the 158 KiB of accumulators allow only two wavefronts per SIMD: with 128 the
// compiler keeps a full eight gathers in flight instead of seven, the wavefronts of a band drift apart and the L2 hit rate
// falls from 53 to 41 % (item step 282 vs 266 us, profiles/r03j_*) -- the same effect as the continuous pipeline of section 8.
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 8))) void k_sweep(SweepArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x / kWave;
    const int w = __builtin_amdgcn_readfirstlane((int)(p.wave_begin + blockIdx.x * (kBlock / kWave) + wib));
    if (w >= p.n_waves) return;
    const int npieces = p.wave_npieces[w];
    if (npieces == 0) return;  // wave-uniform
    const int l = lane & 15, g = lane >> 4;
    const int c0 = min(l * 4, p.dim - 4);
    float *acc = lds + (size_t)wib * (p.row_cap + 1) * 64;     // rows of 64 floats, row `row_cap` = dummy
    {   // zero my region: (row_cap + 1) * 256 B, 1 KiB per wave-instruction
        const f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = lane; i < (p.row_cap + 1) * 16; i += kWave) *reinterpret_cast<f4 *>(acc + i * 4) = z;
    }
    const auto xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, p.x_bytes, 0x00020000);
    const unsigned xs = (unsigned)p.x_stride * 4u, xoff = (unsigned)c0 * 4u;
    float *mine = acc + l * 4;                                  // + piece * 64
    int slab = p.wave_slab_ptr[w];
    const int slab_end = p.wave_slab_ptr[w + 1];
    if (slab >= slab_end) return;
    PieceSlots piece_slots;
    piece_slots.load(p.piece_slot + (int64_t)w * p.row_cap, npieces, lane);
    u4 nxt = __builtin_nontemporal_load(p.slabs + (int64_t)slab * kWave + lane);
#ifdef LGC_SWEEP_TRACE
    int tk = 0;
    if (p.trace && lane == 0) p.trace[(int64_t)w * 16 + tk++] = __builtin_amdgcn_s_memrealtime();
#endif
    for (; slab < slab_end; ++slab) {
        const u4 cur = nxt;
        if (slab + 1 < slab_end) nxt = __builtin_nontemporal_load(p.slabs + (int64_t)(slab + 1) * kWave + lane);
        // 32 steps, DEPTH gathers in flight: issue step s + DEPTH after consuming step s
        f4 xv[DEPTH];
        int pk[DEPTH], vb[DEPTH];
#define LGC_ISSUE(S)                                                                                                   \
    {                                                                                                                  \
        sweep_fetch<((S) >> 1), ((S) & 1)>(cur, pk[(S) % DEPTH], vb[(S) % DEPTH]);                                     \
        xv[(S) % DEPTH] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(                                \
            xsrc, __umul24(pk[(S) % DEPTH] & 0xFFFFFF, xs) + xoff, 0, 0));                          \
    }
#define LGC_CONSUME(S)                                                                                                 \
    {                                                                                                                  \
        float *row = mine + ((unsigned)pk[(S) % DEPTH] >> 24) * 64;                                                    \
        f4 a = *reinterpret_cast<f4 *>(row);                                                                           \
        const float v = __int_as_float(vb[(S) % DEPTH]);                                                               \
        const f2 v2 = {v, v};                                                                                          \
        const f2 lo = {xv[(S) % DEPTH].x, xv[(S) % DEPTH].y}, hi = {xv[(S) % DEPTH].z, xv[(S) % DEPTH].w};             \
        const f2 alo = f2{a.x, a.y} + lo * v2, ahi = f2{a.z, a.w} + hi * v2;                                           \
        *reinterpret_cast<f4 *>(row) = f4{alo.x, alo.y, ahi.x, ahi.y};                                                 \
    }
#define LGC_STEP(S)                                                                                                    \
    LGC_CONSUME(S)                                                                                                     \
    if constexpr ((S) + DEPTH < 32) LGC_ISSUE((S) + DEPTH)
        LGC_ISSUE(0) LGC_ISSUE(1) LGC_ISSUE(2) LGC_ISSUE(3)
        if constexpr (DEPTH >= 8) { LGC_ISSUE(4) LGC_ISSUE(5) LGC_ISSUE(6) LGC_ISSUE(7) }
        LGC_STEP(0) LGC_STEP(1) LGC_STEP(2) LGC_STEP(3) LGC_STEP(4) LGC_STEP(5) LGC_STEP(6) LGC_STEP(7)
        LGC_STEP(8) LGC_STEP(9) LGC_STEP(10) LGC_STEP(11) LGC_STEP(12) LGC_STEP(13) LGC_STEP(14) LGC_STEP(15)
        LGC_STEP(16) LGC_STEP(17) LGC_STEP(18) LGC_STEP(19) LGC_STEP(20) LGC_STEP(21) LGC_STEP(22) LGC_STEP(23)
        LGC_STEP(24) LGC_STEP(25) LGC_STEP(26) LGC_STEP(27) LGC_STEP(28) LGC_STEP(29) LGC_STEP(30) LGC_STEP(31)
#undef LGC_STEP
#undef LGC_CONSUME
#undef LGC_ISSUE
#ifdef LGC_SWEEP_TRACE
        if (p.trace && lane == 0 && tk < 15) p.trace[(int64_t)w * 16 + tk++] = __builtin_amdgcn_s_memrealtime();
#endif
    }
#ifdef LGC_SWEEP_TRACE
    if (p.trace && lane == 0) p.trace[(int64_t)w * 16 + 15] = __builtin_amdgcn_s_memrealtime();   // end of the last slab
#endif
    // write my pieces to their partial slots: lane group g takes pieces g, g + 4, ... (every lane runs the shuffles)
    for (int pc0 = 0; pc0 < npieces; pc0 += 4) {
        const int pc = pc0 + g;
        const int slot = piece_slots.at(min(pc, npieces - 1));
        if (pc < npieces) {
            const f4 a = *reinterpret_cast<const f4 *>(mine + pc * 64);
            __builtin_nontemporal_store(a, reinterpret_cast<f4u *>(p.partials + (int64_t)slot * p.pstride + p.pcol + c0));
        }
    }
}

// One launch per operator half and hop: chunk workgroups first (long rows run longest), then the tile classes.
// Four separate launches cost a rank of an 8-way partition (~130 us of work per hop) a third of its time in launch
// gaps; on one GPU it is a few percent.
struct FusedArgs {
    SpmmArgs sp;
    const lgc_chunk *chunks;
    float *partials;
    TileArgs t[3];
    int32_t n_chunks, chunk_blocks, n_classes;
    int32_t width[3], blocks[3];
};

template <int MODE>   // 0: generic tiles, 1: DPP fast path (61..64 columns), 2: DPP fast path, two DPP rows per row (68..128)
__global__ __launch_bounds__(kBlock) void k_apply_fused(FusedArgs f) {
    int64_t b = blockIdx.x;
    if (b < f.chunk_blocks) {
        chunks_body<4>(f.sp, f.chunks, f.n_chunks, f.partials, b);
        return;
    }
    b -= f.chunk_blocks;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        if (c >= f.n_classes) return;
        if (b < f.blocks[c]) {
            if constexpr (MODE == 2) {
                if (f.width[c] == 8) tiles_body_dpp_wide<8, 1>(f.t[c], b);
                else if (f.width[c] == 16) tiles_body_dpp_wide<16, 1>(f.t[c], b);
                else tiles_body_dpp_wide<32, 2>(f.t[c], b);
            } else if constexpr (MODE == 1) {
                if (f.width[c] == 8) tiles_body_dpp<8, 1>(f.t[c], b);
                else if (f.width[c] == 16) tiles_body_dpp<16, 1>(f.t[c], b);
                else tiles_body_dpp<32, 2>(f.t[c], b);
            } else {
                if (f.width[c] == 8) tiles_body<8, 1>(f.t[c], b);
                else if (f.width[c] == 16) tiles_body<16, 1>(f.t[c], b);
                else tiles_body<32, 2>(f.t[c], b);
            }
            return;
        }
        b -= f.blocks[c];
    }
}

// order + CSR -> tile layout: piece ((t*L + k)*R + s)*PPR + q holds entries k*Wk + 2q, +1 of the row in slot s of tile t
__global__ void k_build_tiles(const int32_t *__restrict__ rowptr, const lgc_entry *__restrict__ entries,
                              const int32_t *__restrict__ order, int64_t n_slots, int32_t W, int32_t L,
                              lgc_entry *__restrict__ slab) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // one thread per slab entry
    if (i >= n_slots * W) return;
    const int R = 128 * L / W, Wk = W / L;
    const int jj = (int)(i % Wk);
    int64_t rest = i / Wk;
    const int s = (int)(rest % R);
    rest /= R;
    const int k = (int)(rest % L);
    const int64_t t = rest / L;
    const int32_t row = order[t * R + s];
    lgc_entry v;
    v.col = -1;
    v.val = 0.0f;
    if (row >= 0) {
        const int32_t b = rowptr[row], e = rowptr[row + 1];
        const int j = k * Wk + jj;
        if (b + j < e) v = entries[b + j];
    }
    slab[i] = v;
}


// The band sweep for tables of 68..96 columns (D = 80, 90): a table row takes two DPP rows (lanes 0-15 columns 0..63,
// lanes 16-31 the rest), a wavefront gathers two rows per instruction, a step has two entries, a slab is 512 bytes
// (both DPP rows of a pair load the same 16 pieces), and an accumulator is 96 floats: 51 per wavefront, five rounds.
constexpr int kWideRow = 96;   // floats per LDS accumulator row

template <int DEPTH>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(1, 8))) void k_sweep_wide(SweepArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & (kWave - 1);
    const int wib = threadIdx.x / kWave;
    const int w = __builtin_amdgcn_readfirstlane((int)(p.wave_begin + blockIdx.x * (kBlock / kWave) + wib));
    if (w >= p.n_waves) return;
    const int npieces = p.wave_npieces[w];
    if (npieces == 0) return;  // wave-uniform
    const int l = lane & 15, pair = lane >> 5, half = (lane >> 4) & 1;
    const int rest = p.dim - 64;                                  // 4 .. 32 columns in the second DPP row
    const bool lane_on = half == 0 || l * 4 < rest;               // lanes that own columns of the table
    const bool lds_on = half == 0 || l < (kWideRow - 64) / 4;     // lanes that own a slice of the accumulator row
    const int c0 = half == 0 ? l * 4 : 64 + min(l * 4, rest - 4);
    float *acc = lds + (size_t)wib * (p.row_cap + 1) * kWideRow;  // row `row_cap` = dummy
    {
        const f4 z = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int i = lane; i < (p.row_cap + 1) * (kWideRow / 4); i += kWave) *reinterpret_cast<f4 *>(acc + i * 4) = z;
    }
    const auto xsrc = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, 0, p.x_bytes, 0x00020000);
    const unsigned xs = (unsigned)p.x_stride * 4u, xoff = (unsigned)c0 * 4u;
    float *mine = acc + half * 64 + (lds_on ? l * 4 : 0);        // + piece * kWideRow
    int slab = p.wave_slab_ptr[w];
    const int slab_end = p.wave_slab_ptr[w + 1];
    if (slab >= slab_end) return;
    PieceSlots piece_slots;
    piece_slots.load(p.piece_slot + (int64_t)w * p.row_cap, npieces, lane);
    // a 512-byte slab = 32 pieces; lanes of pair g read pieces 16 g + l (both of its DPP rows the same ones)
    const int piece_lane = pair * 16 + l;
    u4 nxt = __builtin_nontemporal_load(p.slabs + (int64_t)slab * 32 + piece_lane);
    for (; slab < slab_end; ++slab) {
        const u4 cur = nxt;
        if (slab + 1 < slab_end) nxt = __builtin_nontemporal_load(p.slabs + (int64_t)(slab + 1) * 32 + piece_lane);
        f4 xv[DEPTH];
        int pk[DEPTH], vb[DEPTH];
#define LGC_ISSUE(S)                                                                                                   \
    {                                                                                                                  \
        sweep_fetch<((S) >> 1), ((S) & 1)>(cur, pk[(S) % DEPTH], vb[(S) % DEPTH]);                                     \
        const int col_ = lane_on ? (pk[(S) % DEPTH] & 0xFFFFFF) : 0xFFFFFF;                                            \
        xv[(S) % DEPTH] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(                                \
            xsrc, __umul24(col_, xs) + xoff, 0, 0));                                                \
    }
#define LGC_CONSUME(S)                                                                                                 \
    if (lds_on) {                                                                                                      \
        float *row = mine + ((unsigned)pk[(S) % DEPTH] >> 24) * kWideRow;                                              \
        f4 a = *reinterpret_cast<f4 *>(row);                                                                           \
        const float v = __int_as_float(vb[(S) % DEPTH]);                                                               \
        const f2 v2 = {v, v};                                                                                          \
        const f2 lo = {xv[(S) % DEPTH].x, xv[(S) % DEPTH].y}, hi = {xv[(S) % DEPTH].z, xv[(S) % DEPTH].w};             \
        const f2 alo = f2{a.x, a.y} + lo * v2, ahi = f2{a.z, a.w} + hi * v2;                                           \
        *reinterpret_cast<f4 *>(row) = f4{alo.x, alo.y, ahi.x, ahi.y};                                                 \
    }
#define LGC_STEP(S)                                                                                                    \
    LGC_CONSUME(S)                                                                                                     \
    if constexpr ((S) + DEPTH < 32) LGC_ISSUE((S) + DEPTH)
        LGC_ISSUE(0) LGC_ISSUE(1) LGC_ISSUE(2) LGC_ISSUE(3) LGC_ISSUE(4) LGC_ISSUE(5) LGC_ISSUE(6) LGC_ISSUE(7)
        LGC_STEP(0) LGC_STEP(1) LGC_STEP(2) LGC_STEP(3) LGC_STEP(4) LGC_STEP(5) LGC_STEP(6) LGC_STEP(7)
        LGC_STEP(8) LGC_STEP(9) LGC_STEP(10) LGC_STEP(11) LGC_STEP(12) LGC_STEP(13) LGC_STEP(14) LGC_STEP(15)
        LGC_STEP(16) LGC_STEP(17) LGC_STEP(18) LGC_STEP(19) LGC_STEP(20) LGC_STEP(21) LGC_STEP(22) LGC_STEP(23)
        LGC_STEP(24) LGC_STEP(25) LGC_STEP(26) LGC_STEP(27) LGC_STEP(28) LGC_STEP(29) LGC_STEP(30) LGC_STEP(31)
#undef LGC_STEP
#undef LGC_CONSUME
#undef LGC_ISSUE
    }
    // write my pieces to their partial slots: pair g takes pieces g, g + 2, ... (every lane runs the shuffles)
    for (int pc0 = 0; pc0 < npieces; pc0 += 2) {
        const int pc = pc0 + pair;
        const int slot = piece_slots.at(min(pc, npieces - 1));
        if (pc < npieces && lane_on) {
            const f4 a = *reinterpret_cast<const f4 *>(mine + pc * kWideRow);
            __builtin_nontemporal_store(a, reinterpret_cast<f4u *>(p.partials + (int64_t)slot * p.dim + c0));
        }
    }
}

// Sum of a swept row's partial slots + epilogue, one launch:
//   blocks [0, n_wide)   rows cut into many pieces (hubs): one WORKGROUP per row -- unit (wave, lane group) u of 16
//                        adds slots u, u + 16, ... (4 loads in flight), then groups are added in order inside the
//                        wave (shuffles) and waves in order through LDS: a fixed association;
//   the other blocks     rows with a handful of slots (one per band): one LANE GROUP per row, slots in order --
//                        four rows per wavefront at D=64.
template <int VEC>
__global__ __launch_bounds__(kBlock) void k_sweep_combine(SpmmArgs p, const lgc_multi_row *__restrict__ wide, int32_t n_wide,
                                                         const lgc_multi_row *__restrict__ multi, int32_t n_multi,
                                                         const float *__restrict__ partials) {
    __shared__ float wave_sum[(kBlock / kWave) * 256];
    const int lane = threadIdx.x & (kWave - 1);
    const int wv = threadIdx.x / kWave;
    const int groups = kWave / p.lpr;
    const int g = lane / p.lpr;
    const int l = lane - g * p.lpr;
    const int c0 = lane_column<VEC>(l, p.dim);
    const float *pb = partials + c0;
    if ((int32_t)blockIdx.x < n_wide) {
        const lgc_multi_row mr = wide[blockIdx.x];
        const bool active = g < groups;
        const int units = (kBlock / kWave) * groups;
        Acc<VEC> acc;
        acc.zero();
        if (active) {
            int32_t s = mr.slot_begin + wv * groups + g;
            for (; s + 3 * units < mr.slot_end; s += 4 * units) {
                Acc<VEC> t0 = load_row<VEC>(pb + (int64_t)s * p.dim);
                Acc<VEC> t1 = load_row<VEC>(pb + (int64_t)(s + units) * p.dim);
                Acc<VEC> t2 = load_row<VEC>(pb + (int64_t)(s + 2 * units) * p.dim);
                Acc<VEC> t3 = load_row<VEC>(pb + (int64_t)(s + 3 * units) * p.dim);
#pragma unroll
                for (int i = 0; i < VEC; ++i)
                    acc.v[i] = __fadd_rn(__fadd_rn(__fadd_rn(__fadd_rn(acc.v[i], t0.v[i]), t1.v[i]), t2.v[i]), t3.v[i]);
            }
            for (; s < mr.slot_end; s += units) {
                Acc<VEC> t0 = load_row<VEC>(pb + (int64_t)s * p.dim);
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc.v[i] = __fadd_rn(acc.v[i], t0.v[i]);
            }
        }
        for (int j = 1; j < groups; ++j) {   // every lane executes the shuffles; group 0 keeps the wave's sum
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                float other = __shfl_down(acc.v[i], j * p.lpr);
                if (g == 0) acc.v[i] = __fadd_rn(acc.v[i], other);
            }
        }
        if (active && g == 0) {
#pragma unroll
            for (int i = 0; i < VEC; ++i) wave_sum[wv * 256 + l * VEC + i] = acc.v[i];
        }
        __syncthreads();
        if (wv == 0 && active && g == 0) {
            for (int o = 1; o < kBlock / kWave; ++o) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) acc.v[i] = __fadd_rn(acc.v[i], wave_sum[o * 256 + l * VEC + i]);
            }
            Acc<VEC> rv;
            rv.zero();
            if (p.r != nullptr) rv = load_row<VEC>(p.r + (int64_t)mr.row * p.r_stride + c0);
            finish_row<VEC, SpmmArgs>(p, mr.row, c0, acc, rv);
        }
        return;
    }
    const int64_t m = (((int64_t)blockIdx.x - n_wide) * (kBlock / kWave) + wv) * groups + g;
    if (g >= groups || m >= n_multi) return;
    const lgc_multi_row mr = multi[m];
    Acc<VEC> acc, rv;
    acc.zero();
    rv.zero();
    if (p.r != nullptr) rv = load_row<VEC>(p.r + (int64_t)mr.row * p.r_stride + c0);
    // a row has one slot per band it touches (8 here) plus the cuts of long pieces: eight loads in flight, slots past
    // the row's end read as +0 (adding +0 never changes the sum: it cannot be -0 after its first add), so a row of up to
    // eight slots costs one memory round trip -- it used to be 4 + 4 and then one load at a time for the rest
    for (int32_t s = mr.slot_begin; s < mr.slot_end; s += 8) {
        Acc<VEC> t[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            t[j].zero();
            if (s + j < mr.slot_end) t[j] = load_row<VEC>(pb + (int64_t)(s + j) * p.dim);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int i = 0; i < VEC; ++i) acc.v[i] = __fadd_rn(acc.v[i], t[j].v[i]);
    }
    finish_row<VEC, SpmmArgs>(p, mr.row, c0, acc, rv);
}

// Fixed-order sum of runs: `key` is sorted; position t is a HEAD when key[t] != key[t - 1].  The lane group of a head adds
// vals[t], vals[t + 1], ... of its run in that order (fp32, sequential) and writes y[dest[t]] = (accumulate ? y[dest[t]]
// : 0) + scale * sum; positions that are not heads, and heads with dest < 0, write nothing.  Every destination row is
// owned by one lane group: no atomics, the same bits on every run -- the gradient of a scoring step has a few thousand
// non-zero rows (src/lightgcn.py:123-125 scores 2B pairs), repeated nodes are summed here instead of by float atomics.
__global__ __launch_bounds__(kBlock) void k_segment_sum(const int64_t *__restrict__ key, const int64_t *__restrict__ dest,
                                                       const float *__restrict__ vals, const int32_t *__restrict__ vals_index,
                                                       int64_t n, float scale,
                                                       float *__restrict__ y, int64_t y_stride, int64_t y_rows, int32_t dim,
                                                       int32_t accumulate) {
    const int lane = threadIdx.x & (kWave - 1);
    const int lpr = (dim + 3) / 4, groups = kWave / lpr;
    const int g = lane / lpr, l = lane - g * lpr;
    const int64_t t = ((int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x / kWave)) * groups + g;
    if (g >= groups || t >= n) return;
    const int64_t k = key[t];
    if (t > 0 && key[t - 1] == k) return;
    const int64_t d = dest[t];
    if (d < 0 || d >= y_rows) return;
    const int c0 = l * 4;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int64_t u = t; u < n && key[u] == k; ++u) {
        const int64_t v = vals_index ? (int64_t)vals_index[u] : u;   // the sorted position's row of an UNSORTED value table
        for (int i = 0; i < 4; ++i)
            if (c0 + i < dim) acc[i] = __fadd_rn(acc[i], vals[v * dim + c0 + i]);
    }
    float *out = y + d * y_stride + c0;
    for (int i = 0; i < 4; ++i)
        if (c0 + i < dim) out[i] = __fadd_rn(accumulate ? out[i] : 0.0f, __fmul_rn(scale, acc[i]));
}

// The columns of the listed rows get `value` in `mark` (lgc_seed_mark): the rows the seeded pull has to look at.  Every
// writer stores the same byte, so the order does not matter.
__global__ void k_seed_mark(const int32_t *__restrict__ rowptr, const lgc_entry *__restrict__ entries, int32_t row_begin,
                            int32_t row_end, const int64_t *__restrict__ seed_rows, int64_t n_seed, uint8_t *__restrict__ mark,
                            int64_t mark_len, uint8_t value) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t t = (int64_t)blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;   // one wavefront per listed row
    if (t >= n_seed) return;
    const int64_t row = seed_rows[t];
    if (row < row_begin || row >= row_end) return;
    if (t > 0 && seed_rows[t - 1] == row) return;        // sorted lists: a repeated row is marked once
    for (int32_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += kWave) {
        const int32_t c = entries[k].col;
        if (c >= 0 && c < mark_len) mark[c] = value;
    }
}

// ----------------------------------------------------------------------------------------
// One-time index work of a graph, on the device without torch's index kernels (whose first use in a process loads one
// code object each, lazily: a cold `graph_build_s` was mostly that): the user|item split of an edge list
// (lgc_bipartite_split) and the chunk plan of the long rows (lgc_row_plan_*; graph.build_row_plan is the reference).
// ----------------------------------------------------------------------------------------
__global__ void k_split_init(long long *__restrict__ out) {
    out[0] = -1;                       // max over edges of min(src, dst)
    out[1] = 0x7FFFFFFFFFFFFFFFll;     // min over edges of max(src, dst)
}

__global__ __launch_bounds__(kBlock) void k_bipartite_split(const int64_t *__restrict__ edge_index, int64_t n_edges,
                                                           long long *__restrict__ out) {
    long long lo = -1, hi = 0x7FFFFFFFFFFFFFFFll;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_edges; e += (int64_t)gridDim.x * blockDim.x) {
        const long long a = edge_index[e], b = edge_index[n_edges + e];
        lo = max(lo, min(a, b));
        hi = min(hi, max(a, b));
    }
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) {
        lo = max(lo, (long long)__shfl_xor(lo, off));
        hi = min(hi, (long long)__shfl_xor(hi, off));
    }
    if ((threadIdx.x & (kWave - 1)) == 0) {
        atomicMax(&out[0], lo);
        atomicMin(&out[1], hi);
    }
}

// per row of the range: chunks it is cut into (0 for a row of at most short_max entries), whether it is a multi-chunk
// row, and its partial-sum slots (= chunks, for a multi-chunk row)
__global__ void k_row_plan_count(const int32_t *__restrict__ rowptr, int32_t row_begin, int32_t row_end, int32_t short_max,
                                 int32_t chunk_len, int32_t *__restrict__ nch, int32_t *__restrict__ is_multi,
                                 int32_t *__restrict__ mnch) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_end - row_begin) return;
    const int32_t deg = rowptr[row_begin + i + 1] - rowptr[row_begin + i];
    const int32_t c = deg > short_max ? (deg + chunk_len - 1) / chunk_len : 0;
    nch[i] = c;
    is_multi[i] = c > 1 ? 1 : 0;
    mnch[i] = c > 1 ? c : 0;
}

__global__ void k_row_plan_fill(const int32_t *__restrict__ rowptr, int32_t row_begin, int32_t row_end, int32_t short_max,
                                int32_t chunk_len, const int32_t *__restrict__ chunk_at, const int32_t *__restrict__ multi_at,
                                const int32_t *__restrict__ slot_at, lgc_chunk *__restrict__ chunks,
                                lgc_multi_row *__restrict__ multi) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= row_end - row_begin) return;
    const int32_t row = row_begin + (int32_t)i;
    const int32_t start = rowptr[row], deg = rowptr[row + 1] - start;
    if (deg <= short_max) return;
    const int32_t n = (deg + chunk_len - 1) / chunk_len, per = (deg + n - 1) / n;     // near-equal split
    const int32_t c0 = chunk_at[i];
    for (int32_t j = 0; j < n; ++j) {
        lgc_chunk ch;
        ch.row = row;
        ch.begin = start + j * per;
        ch.end = min(ch.begin + per, start + deg);
        ch.slot = n > 1 ? slot_at[i] + j : -1;
        chunks[c0 + j] = ch;
    }
    if (n > 1) multi[multi_at[i]] = lgc_multi_row{row, slot_at[i], slot_at[i] + n, 0};
}

// ----------------------------------------------------------------------------------------
// Tile classes of the short rows (lgc_tile_classes / lgc_tile_pack): the processing order of lgc_spmm_tiles
// ----------------------------------------------------------------------------------------
// What graph.plan_tile_classes computes with ~25 torch launches (bincount, scatter_reduce, three argsorts, gathers ...),
// as four launches of this library: a process that has never sorted with torch pays 0.3 s for the first use of those
// kernels alone (their code objects are loaded lazily), which is what a cold `plan_build_s` mostly consisted of.
//   key of a row  = (class << 58) | (popularity of its least-gathered column << 31) | that column   (cold order)
//                   class 0 / 1 / 2 = at most 8 / 16 / 32 entries (capped by max_len), 3 = not a tiled row
//   one stable radix sort of (key, row) puts every class in its processing order, ties in row order.
__global__ void k_tile_pop(const int32_t *__restrict__ rowptr, const lgc_entry *__restrict__ entries, int32_t row_begin,
                           int32_t row_end, int32_t *__restrict__ pop) {
    const int64_t k = (int64_t)rowptr[row_begin] + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k < rowptr[row_end]) atomicAdd(&pop[entries[k].col], 1);
}

__global__ void k_tile_keys(const int32_t *__restrict__ rowptr, const lgc_entry *__restrict__ entries, int32_t row_begin,
                            int32_t row_end, int32_t cap, int32_t cold, const int32_t *__restrict__ pop,
                            unsigned long long *__restrict__ keys, int32_t *__restrict__ rows,
                            unsigned long long *__restrict__ class_count) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = i < row_end - row_begin;
    unsigned cls = 4u;
    if (valid) {
        const int32_t row = row_begin + (int32_t)i;
        const int32_t s = rowptr[row], e = rowptr[row + 1], deg = e - s;
        const int c8 = min(8, cap), c16 = min(16, cap), c32 = min(32, cap);
        cls = deg <= c8 ? 0u : deg <= c16 ? 1u : deg <= c32 ? 2u : 3u;
        unsigned long long low = (1ull << 58) - 1;                  // a row without entries sorts last in its class
        if (cold && cls < 3u) {
            for (int32_t k = s; k < e; ++k) {
                const int32_t c = entries[k].col;
                const unsigned long long v = ((unsigned long long)(unsigned)pop[c] << 31) | (unsigned)c;
                low = v < low ? v : low;
            }
        } else if (!cold) {
            low = 0;                                                // natural order: every tie, i.e. row order
        }
        keys[i] = ((unsigned long long)cls << 58) | low;
        rows[i] = row;
    }
    // one atomic per wavefront and class (one per row put 1.6 M 64-bit atomics on four addresses: 16 ms of a 47 ms plan)
    for (unsigned c = 0; c < 4u; ++c) {
        const unsigned long long m = __ballot(cls == c);
        if (m != 0 && (threadIdx.x & (kWave - 1)) == (unsigned)__builtin_ctzll(m)) atomicAdd(&class_count[c], (unsigned long long)__popcll(m));
    }
}

// One thread per tile of a class: its R rows (the class's sorted list, padded with -1), longest first (stable), rank rho
// goes to slot (rho % 4) * B + rho / 4 -- lane group rho % 4, batch rho / 4 --, meta byte bt = longest row of batch bt.
__global__ void k_tile_pack(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ sorted_rows, int64_t n_rows,
                            int32_t R, int64_t n_tiles, int32_t *__restrict__ order, int32_t *__restrict__ meta) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    int32_t row[16], deg[16];
    for (int s = 0; s < R; ++s) {
        const int64_t at = t * R + s;
        row[s] = at < n_rows ? sorted_rows[at] : -1;
        deg[s] = row[s] >= 0 ? rowptr[row[s] + 1] - rowptr[row[s]] : -1;
    }
    for (int a = 1; a < R; ++a) {                                // stable insertion sort, descending by degree
        const int32_t r = row[a], d = deg[a];
        int b = a - 1;
        while (b >= 0 && deg[b] < d) { row[b + 1] = row[b]; deg[b + 1] = deg[b]; --b; }
        row[b + 1] = r;
        deg[b + 1] = d;
    }
    const int B = R / 4;
    int32_t m = 0;
    for (int rho = 0; rho < R; ++rho) {
        order[t * R + (rho % 4) * B + rho / 4] = row[rho];
        if (rho % 4 == 0) m |= max(deg[rho], 0) << (8 * (rho / 4));   // ranks 4 bt .. 4 bt + 3: the first is the longest
    }
    meta[t] = m;
}

// ----------------------------------------------------------------------------------------
// Seed preparation of the sparse backward pass (lgc_seed_prepare): ONE workgroup sorts up to kSeedMax row ids and
// derives everything the segment sums and the seeded pull need -- what the host code did with ~25 small launches
// (sort, gathers, compares, index_puts) per training step.
// ----------------------------------------------------------------------------------------
constexpr int kSeedMax = 8192;
constexpr int kSeedBlock = 1024;

// Pass 1 (m / 8 workgroups): every workgroup keeps all m keys -- row + 1, ids outside the table as "no row" = 0; the
// position is the tie-break -- in LDS and RANKS eight of them by counting the smaller ones, 32 threads per key, each
// scanning 1/32 of the array (LDS reads: a wavefront reads two addresses, both broadcasts); (key, position) pairs are
// distinct, so the ranks are a permutation and sorted[rank] = key is a stable sort by row.  The scan is a dependent chain of
// LDS reads, so its time falls with the threads per key: 4 -> 36.7 us, 8 -> 19.5 us at m = 4096.  A one-workgroup bitonic sort of
// the same keys took 38-51 us (78 barrier stages of LDS-bound 64-bit compare-exchanges); 1024 keys per workgroup with
// one thread per key 69 us (four workgroups on the whole chip).
constexpr int kRankKeys = 8;      // keys ranked per workgroup of 256 threads: 32 threads per key (19.5 us with 8, 36.7 with 4)

__global__ __launch_bounds__(kBlock) void k_seed_rank(const int64_t *__restrict__ rows, int32_t m, int64_t n_nodes,
                                                     unsigned long long *__restrict__ sorted) {
    __shared__ __attribute__((aligned(16))) uint32_t key[kSeedMax];   // row + 1 (0 = "no row"); position = array index: 32 KiB
    __shared__ int part[kBlock];
    for (int i = threadIdx.x; i < m; i += kBlock) {
        int64_t r = rows[i];
        if (r < 0 || r >= n_nodes) r = -1;
        key[i] = (uint32_t)(r + 1);
    }
    __syncthreads();
    constexpr int kParts = kBlock / kRankKeys;
    const int k = threadIdx.x & (kRankKeys - 1), q = threadIdx.x / kRankKeys;     // key k of this workgroup, slice q of the array
    const int i = blockIdx.x * kRankKeys + k;
    const uint32_t mine = i < m ? key[i] : 0u;
    const int per = ((m + 4 * kParts - 1) / (4 * kParts)) * 4, lo = min(m, q * per), hi = min(m, lo + per);   // whole uint4s
    // (row, position) order: everything with a smaller row, and the equal rows in front of me; branch-free, four keys per
    // LDS read, two reads in flight
    int rank = 0;
    int j = lo;
    auto count4 = [&](const u4 o, int at) {
        return (int)(o.x < mine) + (int)((o.x == mine) & (at < i)) + (int)(o.y < mine) + (int)((o.y == mine) & (at + 1 < i)) +
               (int)(o.z < mine) + (int)((o.z == mine) & (at + 2 < i)) + (int)(o.w < mine) + (int)((o.w == mine) & (at + 3 < i));
    };
    for (; j + 8 <= hi; j += 8) {
        const u4 o0 = *reinterpret_cast<const u4 *>(key + j), o1 = *reinterpret_cast<const u4 *>(key + j + 4);
        rank += count4(o0, j) + count4(o1, j + 4);
    }
    for (; j < hi; ++j) {
        const uint32_t o = key[j];
        rank += (int)(o < mine) + (int)((o == mine) & (j < i));
    }
    part[threadIdx.x] = rank;
    __syncthreads();
    if (q == 0 && i < m) {
        int total = 0;
#pragma unroll
        for (int p = 0; p < kParts; ++p) total += part[p * kRankKeys + k];
        sorted[total] = ((unsigned long long)mine << 13) | (unsigned)i;
    }
}

// Pass 2: one thread per sorted position -- run heads, destination lists, the pull's column map.
__global__ void k_seed_finish(const unsigned long long *__restrict__ key, int32_t m, int64_t split,
                              int64_t *__restrict__ rows_sorted, int32_t *__restrict__ perm, int64_t *__restrict__ dest_item,
                              int64_t *__restrict__ dest_slot, int64_t *__restrict__ dest_user,
                              uint8_t *__restrict__ col_flag, int32_t *__restrict__ col_slot) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const unsigned long long k = key[t];
    const int64_t row = (int64_t)(k >> 13) - 1;
    const bool head = t == 0 || (int64_t)(key[t - 1] >> 13) - 1 != row;
    const bool user = row >= 0 && row < split, item = row >= split;
    rows_sorted[t] = row;
    perm[t] = (int32_t)(k & 0x1FFF);
    dest_item[t] = (head && item) ? row : -1;
    dest_slot[t] = (head && user) ? t : -1;
    dest_user[t] = (head && user) ? row : -1;
    if (head && user && col_flag) {
        col_flag[row] = 1;
        col_slot[row] = t;
    }
}

// flag[row] = value for the user rows (0 <= row < split) of a sorted row list: takes the flags of a step's seeds back
__global__ void k_seed_flags(const int64_t *__restrict__ rows_sorted, int64_t m, int64_t split, uint8_t *__restrict__ flag,
                             uint8_t value) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m) return;
    const int64_t row = rows_sorted[t];
    if (row >= 0 && row < split) flag[row] = value;
}

// Pair scoring that keeps what the backward pass needs (lgc_pair_dot_rows): scores as k_pair_dot, plus the two gathered
// rows of every pair and a validity byte -- instead of four compares, three ands, two clamps and two row gathers on the
// host side.  An out-of-range pair scores NaN, keeps zero rows, ok = 0, and raises the status bit.
__global__ __launch_bounds__(kBlock) void k_pair_dot_rows(const float *__restrict__ emb, int64_t stride, int32_t dim,
                                                         int64_t n_nodes, const int64_t *__restrict__ idx0,
                                                         const int64_t *__restrict__ idx1, int64_t n_pairs,
                                                         float *__restrict__ scores, float *__restrict__ rows0,
                                                         float *__restrict__ rows1, uint8_t *__restrict__ ok,
                                                         int32_t *__restrict__ status) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t m = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x / kWave);
    if (m >= n_pairs) return;
    const int64_t a = idx0[m], b = idx1[m];
    const bool valid = a >= 0 && a < n_nodes && b >= 0 && b < n_nodes;   // wave-uniform
    if (!valid && lane == 0) {
        atomicOr(status, LGC_ST_INDEX_OOB);
        scores[m] = NAN;
    }
    if (lane == 0 && ok) ok[m] = valid ? 1 : 0;
    const float *pa = emb + a * stride, *pb = emb + b * stride;
    float s = 0.0f;
    for (int c = lane; c < dim; c += kWave) {
        const float va = valid ? pa[c] : 0.0f, vb = valid ? pb[c] : 0.0f;
        s += va * vb;
        if (rows0) rows0[m * dim + c] = va;
        if (rows1) rows1[m * dim + c] = vb;
    }
    if (!valid) return;
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) scores[m] = s;
}

// The seed of the backward pass from the gradient of the scores (lgc_pair_seed_vals):
//   vals[m]           = g[m] * rows1[m]      d score_m / d out[idx0[m]] = out[idx1[m]]
//   vals[n_pairs + m] = g[m] * rows0[m]      d score_m / d out[idx1[m]] = out[idx0[m]]
// with g[m] = mask[m] ? grad_scores[m] * (*grad_scale) : 0.  grad_scale: an optional DEVICE scalar (the upstream
// gradient of a loss this node computed itself), so that no host sync is needed to read it.
__global__ __launch_bounds__(kBlock) void k_pair_seed_vals(const float *__restrict__ grad_scores, const uint8_t *__restrict__ mask,
                                                          const float *__restrict__ grad_scale, const float *__restrict__ rows0,
                                                          const float *__restrict__ rows1, int64_t n_pairs, int32_t dim,
                                                          float *__restrict__ vals) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs * dim) return;
    const int64_t m = i / dim;
    float g = (mask == nullptr || mask[m]) ? grad_scores[m] : 0.0f;
    if (grad_scale) g = __fmul_rn(g, *grad_scale);
    vals[i] = __fmul_rn(g, rows1[i]);
    vals[n_pairs * dim + i] = __fmul_rn(g, rows0[i]);
}

// BPR loss of one batch of triples and its gradient with respect to the scores (lgc_bpr_loss): what
// `recommendation_loss(out[:B], out[B:], 0) * B` of src/train_lightgcn.py:141 (src/lightgcn.py:262-286 with lambda_reg = 0)
// and its autograd compute with ~15 launches:  loss = -sum_{t: mask[t]} log sigmoid(s[t] - s[B + t]) / size,
// grad[t] = -sigmoid(-(s[t] - s[B + t])) / size, grad[B + t] = -grad[t] (0 where the mask is off).  One workgroup, a fixed
// reduction tree: the same bits on every run.  logsigmoid(d) = min(d, 0) - log1p(exp(-|d|)), torch's formula.
__global__ __launch_bounds__(kSeedBlock) void k_bpr_loss(const float *__restrict__ scores, const uint8_t *__restrict__ mask,
                                                        int64_t n_triples, float inv_size, float *__restrict__ loss,
                                                        float *__restrict__ grad) {
    __shared__ float part[kSeedBlock];
    float acc = 0.0f;
    for (int64_t t = threadIdx.x; t < n_triples; t += kSeedBlock) {
        const bool on = mask == nullptr || mask[t] != 0;
        const float d = on ? scores[t] - scores[n_triples + t] : 0.0f;
        const float ls = fminf(d, 0.0f) - log1pf(expf(-fabsf(d)));
        const float sg = 1.0f / (1.0f + expf(d));                  // sigmoid(-d)
        if (on) acc += ls;
        grad[t] = on ? -sg * inv_size : 0.0f;
        grad[n_triples + t] = on ? sg * inv_size : 0.0f;
    }
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = kSeedBlock / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = -part[0] * inv_size;
}

// The regulariser of src/utils_v2.py:193-211 on three id lists of one table (lgc_reg_rows): what
//   (1/2) * (w[u].norm().pow(2) + w[p].norm().pow(2) + w[n].norm().pow(2)) / size * decay
// costs as 13 torch launches (three gathers into [B, D] copies, three norms, pows, adds, scalings) plus nine more that
// normalise the ids for the gradient's row list -- in ONE workgroup: a thread per row (columns in order), per-list sums of
// squares added up in a fixed order (the same bits on every run),
// value = scale * ((sqrt S_u)^2 + (sqrt S_p)^2 + (sqrt S_n)^2) like the expression above.  rows_out (int64 [m0 + m1 + m2],
// optional): the ids as row numbers, negative ids wrapped (torch's indexing), an id outside [-n_rows, n_rows) as -1 = "no
// row" -- such an id contributes nothing and sets LGC_ST_INDEX_OOB (upstream's gather raises).
__global__ __launch_bounds__(kSeedBlock) void k_reg_rows(const float *__restrict__ w, int64_t stride, int32_t dim, int64_t n_rows,
                                                        const int64_t *__restrict__ ids0, int64_t m0,
                                                        const int64_t *__restrict__ ids1, int64_t m1,
                                                        const int64_t *__restrict__ ids2, int64_t m2, float scale,
                                                        float *__restrict__ value, int64_t *__restrict__ rows_out,
                                                        int32_t *__restrict__ status) {
    constexpr int kWaves = kSeedBlock / kWave;
    __shared__ float part[3][kWaves];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    const int64_t total = m0 + m1 + m2;
    float acc[3] = {0.0f, 0.0f, 0.0f};
    // a THREAD per row (a wavefront per row made one workgroup walk 3 B rows through two dependent loads each: 200 us): the
    // dim / 4 loads of a row are independent, a thread has all of them in flight
    for (int64_t t = threadIdx.x; t < total; t += kSeedBlock) {
        const int which = t < m0 ? 0 : (t < m0 + m1 ? 1 : 2);
        int64_t id = which == 0 ? ids0[t] : (which == 1 ? ids1[t - m0] : ids2[t - m0 - m1]);
        if (id < 0) id += n_rows;
        const bool ok = id >= 0 && id < n_rows;
        if (rows_out != nullptr) rows_out[t] = ok ? id : -1;
        if (!ok) {
            atomicOr(status, LGC_ST_INDEX_OOB);
            continue;
        }
        const float *row = w + id * stride;
        float sq = 0.0f;
        int c = 0;
        for (; c + 4 <= dim; c += 4) {
            const f4 v = *reinterpret_cast<const f4u *>(row + c);
            sq += v.x * v.x;
            sq += v.y * v.y;
            sq += v.z * v.z;
            sq += v.w * v.w;
        }
        for (; c < dim; ++c) sq += row[c] * row[c];
        acc[which] += sq;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float v = acc[j];
        for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off);
        if (lane == 0) part[j][wv] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float out = 0.0f;
        for (int j = 0; j < 3; ++j) {
            float sum = 0.0f;
            for (int i = 0; i < kWaves; ++i) sum += part[j][i];
            const float nrm = sqrtf(sum);
            out += nrm * nrm;
        }
        value[0] = out * scale;
    }
}

struct LincombArgs {
    const float *src[LGC_MAX_TERMS];
    int64_t stride[LGC_MAX_TERMS];
    float coef[LGC_MAX_TERMS];
    int32_t n_terms;
};

__global__ void k_lincomb(float *__restrict__ y, int64_t y_stride, LincombArgs a, int64_t n_rows, int32_t dim) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t total = n_rows * dim;
    for (; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / dim;
        const int32_t c = (int32_t)(i - row * dim);
        float v = __fmul_rn(a.coef[0], a.src[0][row * a.stride[0] + c]);
        for (int t = 1; t < a.n_terms; ++t) v = __fadd_rn(v, __fmul_rn(a.coef[t], a.src[t][row * a.stride[t] + c]));
        y[row * y_stride + c] = v;
    }
}

// ----------------------------------------------------------------------------------------
// Dense Adam step over the embedding table (the caller's optimizer.step(), src/train_lightgcn.py:58,147)
// ----------------------------------------------------------------------------------------
// One pass: w, g, m, v read once, w, m, v written once (7 x 434 MB at 1.7 M x 64): torch.optim.Adam's arithmetic for
// amsgrad=False, weight_decay=0, maximize=False --
//   m <- m + (g - m) (1 - beta1);  v <- beta2 v + (1 - beta2) g g;  w <- w - (lr / bc1) m / (sqrt(v) / sqrt(bc2) + eps)
// with bc1 = 1 - beta1^t, bc2 = 1 - beta2^t computed by the host in double and handed over as step_size, bc2_sqrt.
// (1 - beta1) and (1 - beta2) come from the host, rounded from double like torch's scalars: 1.0f - 0.999f is 4.7e-5 off.
constexpr int kAdamU = 2;
__global__ __launch_bounds__(kBlock) void k_adam(float *__restrict__ w, const float *__restrict__ g, float *__restrict__ m,
                                                float *__restrict__ v, int64_t n4, int64_t n, float beta2, float omb1, float omb2,
                                                float eps, float step_size, float bc2_sqrt, const float *__restrict__ hyper) {
    if (hyper != nullptr) {   // lgc_adam_step_hp: the step's scalars live in device memory (a captured launch is replayed with
        omb1 = hyper[0]; beta2 = hyper[1]; omb2 = hyper[2]; eps = hyper[3]; step_size = hyper[4]; bc2_sqrt = hyper[5];   // new values)
    }
    auto one = [&](float &wi, float gi, float &mi, float &vi) {
        mi = mi + (gi - mi) * omb1;
        vi = beta2 * vi + omb2 * gi * gi;
        wi = wi - step_size * (mi / (sqrtf(vi) / bc2_sqrt + eps));
    };
    constexpr int U = kAdamU;                 // float4s per thread and array: 8 loads in flight per thread (1 / 4: the same time)
    const int64_t base = ((int64_t)blockIdx.x * blockDim.x) * U + threadIdx.x;
    f4 w4[U], g4[U], m4[U], v4[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = base + (int64_t)u * blockDim.x;
        if (i < n4) {
            w4[u] = reinterpret_cast<f4 *>(w)[i];
            g4[u] = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(g) + i);   // read once: 566 -> 546 us for 108 M elements
            m4[u] = reinterpret_cast<f4 *>(m)[i];
            v4[u] = reinterpret_cast<f4 *>(v)[i];
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int64_t i = base + (int64_t)u * blockDim.x;
        if (i < n4) {
            float wv[4] = {w4[u].x, w4[u].y, w4[u].z, w4[u].w}, gv[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w};
            float mv[4] = {m4[u].x, m4[u].y, m4[u].z, m4[u].w}, vv[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) one(wv[j], gv[j], mv[j], vv[j]);
            reinterpret_cast<f4 *>(w)[i] = f4{wv[0], wv[1], wv[2], wv[3]};
            reinterpret_cast<f4 *>(m)[i] = f4{mv[0], mv[1], mv[2], mv[3]};
            reinterpret_cast<f4 *>(v)[i] = f4{vv[0], vv[1], vv[2], vv[3]};
        }
    }
    // tail (n not a multiple of 4): the first threads of block 0
    const int64_t t = n4 * 4 + threadIdx.x;
    if (blockIdx.x == 0 && t < n) one(w[t], g[t], m[t], v[t]);
}

// ----------------------------------------------------------------------------------------
// Pair scoring
// ----------------------------------------------------------------------------------------
// One wavefront per pair; lanes stride the feature axis, butterfly reduce over 64 lanes.
__global__ __launch_bounds__(kBlock) void k_pair_dot(const float *__restrict__ emb, int64_t stride, int32_t dim,
                                                    int64_t n_nodes, const int64_t *__restrict__ idx0,
                                                    const int64_t *__restrict__ idx1, int64_t n_pairs,
                                                    float *__restrict__ scores, int32_t *__restrict__ status) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t m = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x / kWave);
    if (m >= n_pairs) return;
    const int64_t a = idx0[m], b = idx1[m];
    if (a < 0 || a >= n_nodes || b < 0 || b >= n_nodes) {  // wave-uniform
        if (lane == 0) {
            atomicOr(status, LGC_ST_INDEX_OOB);
            scores[m] = NAN;
        }
        return;
    }
    const float *pa = emb + a * stride, *pb = emb + b * stride;
    float s = 0.0f;
    for (int c = lane; c < dim; c += kWave) s += pa[c] * pb[c];
#pragma unroll
    for (int off = kWave / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) scores[m] = s;
}

// ----------------------------------------------------------------------------------------
// Serving tail: multiplicative seen-mask + top-k per row, on the device
// ----------------------------------------------------------------------------------------
// masked[i] = score[i] * (1 - seen[i])  (src/lightgcn.py:175 -- seen items become 0, they are not removed), then the
// k largest by (value descending, index ascending).  One workgroup per row.  Rows of up to 65,536 columns are read
// ONCE: each thread keeps its 64 order-preserving keys in registers.  Short cut: the k-th largest of the 1,024
// per-thread maxima bounds the k-th largest element from below; the few elements in or above its 11-bit bin go to a
// list in LDS and each counts the entries ahead of it (= its output position).  Heavily tied or flat rows (list
// longer than 512) and wider rows take the general path: three radix passes (11 + 11 + 10 bits, histograms in LDS)
// find the k-th largest key T, one pass collects everything above T plus as many elements equal to T as are still
// needed -- lowest indices first -- and a bitonic sort orders the k winners.  Nothing but [rows, k] leaves the device.
constexpr int kTopkMax = 256;
constexpr int kTopkBlock = 1024;   // 16 wavefronts on one row: a single-row request is latency-bound on one CU

__device__ __forceinline__ uint32_t order_key(float v) {
    const uint32_t u = __float_as_uint(__fadd_rn(v, 0.0f));   // -0 -> +0: they compare equal (a seen item's 0 * score)
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);        // ascending with the value; +NaN above +inf like torch.topk
}

constexpr int kTopkRegs = 64;      // keys a thread can hold: rows up to kTopkRegs * kTopkBlock columns are read once
constexpr int kTopkCopies = 4;     // histogram copies (lane & 3), one bank apart: scores crowd into a few exponent bins
constexpr int kTopkHistStride = 2049;
constexpr int kTopkShort = 512;    // longest candidate list the short cut ranks by counting

// REGS: the row's masked keys live in registers (one read of the row, all passes on registers); otherwise every pass
// streams the row again (rows wider than kTopkRegs * kTopkBlock columns).
#ifdef LGC_TOPK_TRACE   // debug builds (tools/topk_trace.py): phase time stamps of block 0
__device__ unsigned long long g_topk_trace[16];
#define LGC_TOPK_STAMP(slot) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_topk_trace[slot] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define LGC_TOPK_STAMP(slot) do { } while (0)
#endif

// MASK: 0 none, 1 dense [rows, n_cols] floats, 2 per-user item lists.
template <bool REGS, int MASK>
__global__ __launch_bounds__(kTopkBlock) void k_mask_topk(const float *__restrict__ scores, int64_t score_stride,
                                                         const float *__restrict__ seen, int64_t seen_stride,
                                                         const int64_t *__restrict__ list_ptr,
                                                         const int64_t *__restrict__ list_items,
                                                         const int64_t *__restrict__ list_rows, int32_t n_cols,
                                                         int32_t k, int64_t *__restrict__ out_index,
                                                         float *__restrict__ out_value) {
    extern __shared__ uint32_t seen_bits[];     // list form of the mask: one bit per column, built here
    __shared__ uint32_t hist[kTopkCopies * kTopkHistStride];
    __shared__ uint32_t cand_key[kTopkMax], cand_inv[kTopkMax];   // candidates: key, then ~index (lowest index wins ties)
    __shared__ uint32_t sh_bin, sh_need, sh_count, sh_short, sh_wave[kTopkBlock / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wv = tid / kWave;
    const float *srow = scores + (int64_t)blockIdx.x * score_stride;
    const float *mrow = MASK == 1 ? seen + (int64_t)blockIdx.x * seen_stride : nullptr;
    LGC_TOPK_STAMP(0);
    if (MASK == 2) {   // seen items of this row's user as a bitmask in LDS: the dense [rows, n_cols] mask never exists
        for (int b = tid; b < (n_cols + 31) / 32; b += kTopkBlock) seen_bits[b] = 0u;
        __syncthreads();
        const int64_t u = list_rows ? list_rows[blockIdx.x] : (int64_t)blockIdx.x;
        for (int64_t e = list_ptr[u] + tid; e < list_ptr[u + 1]; e += kTopkBlock) {
            const int64_t it = list_items[e];
            if (it >= 0 && it < n_cols) atomicOr(&seen_bits[it >> 5], 1u << (it & 31));
        }
        __syncthreads();
    }
    // score * (1 - seen) in upstream's arithmetic; the list form has seen = 1 for listed columns, 0 elsewhere
    auto masked_at = [&](float s, float m, int i) {
        if (MASK == 2) return (seen_bits[i >> 5] >> (i & 31)) & 1u ? __fmul_rn(s, 0.0f) : s;
        return MASK == 1 ? __fmul_rn(s, __fsub_rn(1.0f, m)) : s;
    };
    // slots past the row end hold key 0, below every real key (real keys are lifted to >= 1: only the one -NaN
    // pattern 0xFFFFFFFF moves, onto its neighbour), so the passes need no bounds test.  Loads are clamped, not
    // predicated: no divergent control flow around them.
    auto key_of = [&](float s, float m, int i) {
        const uint32_t key = max(order_key(masked_at(s, m, min(i, n_cols - 1))), 1u);
        return i < n_cols ? key : 0u;
    };
    uint32_t keys[REGS ? kTopkRegs : 1];
    if (REGS) {
        constexpr int G = 16;                         // independent loads per thread in flight: one CU, latency-bound
#pragma unroll
        for (int j0 = 0; j0 < kTopkRegs; j0 += G) {
            if (j0 * kTopkBlock < n_cols) {           // block-uniform
                float sv[G], mv[G];
#pragma unroll
                for (int j = 0; j < G; ++j) {
                    const int i = min((j0 + j) * kTopkBlock + tid, n_cols - 1);
                    sv[j] = srow[i];
                    mv[j] = MASK == 1 ? mrow[i] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < G; ++j) keys[j0 + j] = key_of(sv[j], mv[j], (j0 + j) * kTopkBlock + tid);
            } else {
#pragma unroll
                for (int j = 0; j < G; ++j) keys[j0 + j] = 0u;
            }
        }
    }
    LGC_TOPK_STAMP(1);
    uint32_t *my_hist = hist + (lane & (kTopkCopies - 1)) * kTopkHistStride;
    // Walk the nb bins of hist[] down from the top until `want` elements are covered: sh_bin = the bin that crosses,
    // sh_need = elements still wanted from it, sh_count = its population.  Thread t owns nb / 1024 bins (block scan).
    auto find_bin = [&](int nb, uint32_t want) {
        const int per = nb / kTopkBlock;                     // 2 or 1
        uint32_t mine = 0;
        for (int j = 0; j < per; ++j) mine += hist[nb - 1 - (tid * per + j)];
        uint32_t incl = mine;                                // inclusive scan, thread 0 = highest bins
        for (int off = 1; off < kWave; off <<= 1) {
            const uint32_t o = __shfl_up(incl, off);
            if (lane >= off) incl += o;
        }
        if (lane == kWave - 1) sh_wave[wv] = incl;
        __syncthreads();
        for (int q = 0; q < wv; ++q) incl += sh_wave[q];
        const uint32_t before = incl - mine;
        if (before < want && incl >= want) {                 // exactly one thread
            uint32_t acc = before;
            for (int j = 0; j < per; ++j) {
                const int b = nb - 1 - (tid * per + j);
                if (acc + hist[b] >= want) { sh_bin = (uint32_t)b; sh_need = want - acc; sh_count = hist[b]; break; }
                acc += hist[b];
            }
        }
        __syncthreads();
    };
    auto fold_copies = [&](int nb) {
        for (int b = tid; b < nb; b += kTopkBlock) {
            uint32_t c = hist[b];
#pragma unroll
            for (int q = 1; q < kTopkCopies; ++q) c += hist[q * kTopkHistStride + b];
            hist[b] = c;
        }
        __syncthreads();
    };
    // Short cut (keys in registers): the k-th largest of the 1024 per-thread maxima is a lower bound of the k-th
    // largest element, and on anything but heavily tied rows only a few dozen elements reach its 11-bit bin.  Those
    // go to a short list in LDS and every entry counts the entries ahead of it: its count is its output position.
    // More than kTopkShort entries (ties, flat rows): the general radix passes below do the row.
    if (REGS) {
        uint32_t mx = 0;
#pragma unroll
        for (int j = 0; j < kTopkRegs; ++j) mx = max(mx, keys[j]);
        for (int b = tid; b < kTopkCopies * kTopkHistStride; b += kTopkBlock) hist[b] = 0;
        if (tid == 0) sh_short = 0;
        __syncthreads();
        atomicAdd(&my_hist[mx >> 21], 1u);                   // a thread without a column has mx = 0: bin 0, never needed
        __syncthreads();
        fold_copies(2048);
        find_bin(2048, (uint32_t)k);
        const uint32_t low = max(sh_bin << 21, 1u);
        uint32_t *short_key = hist, *short_inv = hist + kTopkShort;   // the histogram is free again
#pragma unroll
        for (int j = 0; j < kTopkRegs; ++j)
            if (j * kTopkBlock < n_cols && keys[j] >= low) {
                const uint32_t pos = atomicAdd(&sh_short, 1u);
                if (pos < kTopkShort) {
                    short_key[pos] = keys[j];
                    short_inv[pos] = 0xFFFFFFFFu - (uint32_t)(j * kTopkBlock + tid);
                }
            }
        __syncthreads();
        LGC_TOPK_STAMP(2);
        const uint32_t n_short = sh_short;
        if (n_short <= kTopkShort) {
            if (wv * kWave < (int)n_short) {                 // whole wavefronts past the list have nothing to rank
                const uint32_t kt = tid < (int)n_short ? short_key[tid] : 0u, it = tid < (int)n_short ? short_inv[tid] : 0u;
                uint32_t ahead = 0;
                for (uint32_t c0 = 0; c0 < n_short; c0 += 8) {   // broadcast reads, eight entries in flight; slots past
                    uint32_t kc[8], ic[8];                       // the list end read as 0 = behind every entry
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const bool in = c0 + q < n_short;
                        kc[q] = in ? short_key[c0 + q] : 0u;
                        ic[q] = in ? short_inv[c0 + q] : 0u;
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) ahead += (kc[q] > kt || (kc[q] == kt && ic[q] > it)) ? 1u : 0u;
                }
                if (tid < (int)n_short && ahead < (uint32_t)k) {
                    const uint32_t idx = 0xFFFFFFFFu - it;
                    out_index[(int64_t)blockIdx.x * k + ahead] = (int64_t)idx;
                    if (out_value)
                        out_value[(int64_t)blockIdx.x * k + ahead] =
                            masked_at(srow[idx], MASK == 1 ? mrow[idx] : 0.0f, (int)idx);
                }
            }
            LGC_TOPK_STAMP(12);
            return;
        }
        __syncthreads();
    }
    uint32_t prefix = 0, mask = 0, need = (uint32_t)k, eq_total = 0;
    const int shifts[3] = {21, 10, 0}, bits[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = shifts[pass], nb = 1 << bits[pass];
        for (int b = tid; b < kTopkCopies * kTopkHistStride; b += kTopkBlock) hist[b] = 0;
        __syncthreads();
        if (REGS) {
#pragma unroll
            for (int j = 0; j < kTopkRegs; ++j)
                if (j * kTopkBlock < n_cols && (keys[j] & mask) == prefix)
                    atomicAdd(&my_hist[(keys[j] >> shift) & (nb - 1)], 1u);
        } else {
            for (int base = 0; base < n_cols; base += 4 * kTopkBlock) {   // four independent loads per thread in flight
                float sv[4], mv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = min(base + j * kTopkBlock + tid, n_cols - 1);
                    sv[j] = srow[i];
                    mv[j] = MASK == 1 ? mrow[i] : 0.0f;
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = base + j * kTopkBlock + tid;
                    const uint32_t key = key_of(sv[j], mv[j], i);
                    if (i < n_cols && (key & mask) == prefix) atomicAdd(&my_hist[(key >> shift) & (nb - 1)], 1u);
                }
            }
        }
        __syncthreads();
        LGC_TOPK_STAMP(2 + 3 * pass);
        fold_copies(nb);                               // into copy 0
        LGC_TOPK_STAMP(3 + 3 * pass);
        find_bin(nb, need);
        prefix |= sh_bin << shift;
        mask |= (uint32_t)(nb - 1) << shift;
        need = sh_need;
        eq_total = sh_count;
        __syncthreads();
        LGC_TOPK_STAMP(4 + 3 * pass);
    }
    const uint32_t T = prefix, n_gt = (uint32_t)k - need;   // take all keys > T (n_gt of them) and `need` keys == T
    if (tid == 0) sh_count = 0;
    for (int i = tid; i < kTopkMax; i += kTopkBlock) cand_key[i] = cand_inv[i] = 0u;
    __syncthreads();
    const bool ties_cut = eq_total > need;                  // more elements equal T than fit: lowest indices win
    uint32_t eq_taken = 0;                                  // block-uniform, only used when ties_cut
    auto collect = [&](int i, uint32_t key) {   // key 0 marks a slot past the row end (T >= 1 whenever k <= n_cols)
        const bool gt = key > T, eq = key == T;
        if (gt || (eq && !ties_cut)) {
            const uint32_t pos = atomicAdd(&sh_count, 1u);
            cand_key[pos] = key;
            cand_inv[pos] = 0xFFFFFFFFu - (uint32_t)i;
        }
        if (ties_cut && eq_taken < need) {   // ordered by index: ballots + per-wave offsets (rare: exact ties at the cut)
            const unsigned long long bal = __ballot(eq);
            if (lane == 0) sh_wave[wv] = (uint32_t)__popcll(bal);
            __syncthreads();
            uint32_t off = eq_taken, tot = 0;
            for (int q = 0; q < kTopkBlock / kWave; ++q) { if (q < wv) off += sh_wave[q]; tot += sh_wave[q]; }
            const uint32_t rank = off + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
            if (eq && rank < need) {
                cand_key[n_gt + rank] = key;
                cand_inv[n_gt + rank] = 0xFFFFFFFFu - (uint32_t)i;
            }
            eq_taken += tot;
            __syncthreads();
        }
    };
    if (REGS) {
#pragma unroll
        for (int j = 0; j < kTopkRegs; ++j)
            if (j * kTopkBlock < n_cols) collect(j * kTopkBlock + tid, keys[j]);   // block-uniform condition
    } else {
        for (int base = 0; base < n_cols; base += kTopkBlock) {
            const int i = base + tid;
            const int ic = min(i, n_cols - 1);
            collect(i, key_of(srow[ic], MASK == 1 ? mrow[ic] : 0.0f, i));
        }
    }
    __syncthreads();
    LGC_TOPK_STAMP(11);
    // bitonic sort, descending, of the first n_sort >= k candidate slots (unused slots are 0 = below every real key)
    int n_sort = 2;
    while (n_sort < k) n_sort <<= 1;
    for (int size = 2; size <= n_sort; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int a = tid, b = tid ^ stride;
            if (a < n_sort && b > a) {
                const bool desc = (a & size) == 0;
                const uint32_t xk = cand_key[a], yk = cand_key[b], xi = cand_inv[a], yi = cand_inv[b];
                const bool x_lt_y = xk < yk || (xk == yk && xi < yi);
                const bool differ = xk != yk || xi != yi;
                if (differ && (desc ? x_lt_y : !x_lt_y)) {
                    cand_key[a] = yk; cand_inv[a] = yi;
                    cand_key[b] = xk; cand_inv[b] = xi;
                }
            }
            __syncthreads();
        }
    }
    LGC_TOPK_STAMP(12);
    if (tid < k) {
        const uint32_t idx = 0xFFFFFFFFu - cand_inv[tid];
        out_index[(int64_t)blockIdx.x * k + tid] = (int64_t)idx;
        if (out_value)
            out_value[(int64_t)blockIdx.x * k + tid] = masked_at(srow[idx], MASK == 1 ? mrow[idx] : 0.0f, (int)idx);
    }
}

// ----------------------------------------------------------------------------------------
// Mini-batch sampler
// ----------------------------------------------------------------------------------------
// splitmix64 finaliser as a counter-based generator: draw(seed, step, sample, attempt) is stateless.
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// unbiased integer in [0, range): 64-bit multiply-high of a 64-bit draw (bias < range / 2^64)
__device__ __forceinline__ uint64_t bounded(uint64_t r, uint64_t range) { return __umul64hi(r, range); }

__global__ void k_sample_triples(const int64_t *__restrict__ users, int64_t n, const int32_t *__restrict__ pos_ptr,
                                 const int64_t *__restrict__ pos_items, const int32_t *__restrict__ ign_ptr,
                                 const int64_t *__restrict__ ign_items, int64_t n_users, int64_t n_items,
                                 uint64_t seed, uint64_t step, int64_t *__restrict__ pos_out,
                                 int64_t *__restrict__ neg_out, int32_t *__restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int64_t u = users[i];
    const uint64_t key = mix64(mix64(seed) ^ mix64(step * 0xD1B54A32D192ED03ull + (uint64_t)i));
    if (u < 0 || u >= n_users || pos_ptr[u + 1] == pos_ptr[u]) {
        atomicOr(status, LGC_ST_INDEX_OOB);
        pos_out[i] = neg_out[i] = n_users;
        return;
    }
    const int32_t pb = pos_ptr[u], pc = pos_ptr[u + 1] - pb;
    pos_out[i] = pos_items[pb + (int64_t)bounded(mix64(key), (uint64_t)pc)];
    const int32_t ib = ign_ptr[u], ie = ign_ptr[u + 1];
    int64_t cand = n_users;
    bool ok = false;
    for (int attempt = 0; attempt < 256 && !ok; ++attempt) {
        cand = n_users + (int64_t)bounded(mix64(key + 0x632BE59BD9B4E019ull * (uint64_t)(attempt + 1)), (uint64_t)n_items);
        int32_t lo = ib, hi = ie;                       // binary search in the sorted ignore set
        while (lo < hi) {
            const int32_t mid = lo + ((hi - lo) >> 1);
            if (ign_items[mid] < cand) lo = mid + 1; else hi = mid;
        }
        ok = !(lo < ie && ign_items[lo] == cand);
    }
    if (!ok) atomicOr(status, LGC_ST_SAMPLER_EXHAUSTED);
    neg_out[i] = cand;
}

// ----------------------------------------------------------------------------------------
// dispatch helpers
// ----------------------------------------------------------------------------------------
struct DimCfg {
    int vec;  // 4 (any D >= 4, the last lane overlapping when D % 4 != 0) or 1 (D < 4)
    int lpr;
};

bool dim_cfg(int32_t dim, DimCfg *cfg) {
    if (dim < 1 || dim > 256) return false;
    cfg->vec = dim >= 4 ? 4 : 1;
    cfg->lpr = dim >= 4 ? (dim + 3) / 4 : dim;
    return true;
}

template <typename F>
int dispatch_dim(const DimCfg &cfg, F &&f) {
    if (cfg.vec == 4) return f(std::integral_constant<int, 4>{});
    return f(std::integral_constant<int, 1>{});
}

bool aligned_to(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace


// ----------------------------------------------------------------------------------------
// Band-sweep planner (host code, host pointers): see k_sweep
// ----------------------------------------------------------------------------------------
struct lgc_sweep_dplan {          // host handle of the device planner between _create and _fill
    lgc_sweep_cfg cfg{};
    lgc_sweep_dims dims{};
    int32_t row_begin = 0;
    std::vector<lgc_multi_row> multi;
    std::vector<int32_t> wave_npieces, piece_slot, wave_item_ptr, wave_slab_ptr;
    // device pointers into the caller's workspace (must stay alive until _fill has returned)
    const unsigned long long *keys = nullptr;
    const int32_t *idx = nullptr, *d_item_ptr = nullptr;
    const lgc_entry *sorted = nullptr;
    int32_t *d_slab_ptr = nullptr;
};

struct lgc_sweep_plan {
    lgc_sweep_dims dims{};
    std::vector<uint32_t> slabs;          // n_slabs * 256 dwords
    std::vector<int32_t> wave_slab_ptr;   // n_waves + 1
    std::vector<int32_t> wave_npieces;    // n_waves
    std::vector<int32_t> piece_slot;      // n_waves * row_cap
    std::vector<lgc_multi_row> multi;     // one per row of the range
};

namespace {

struct SweepPiece {
    int32_t begin, count;   // into the column-sorted entry array
    int32_t band;
};

template <class F>
void parallel_for(int64_t n, F &&f) {
    unsigned nt = std::thread::hardware_concurrency();
    if (const char *e = getenv("LGCN_PLAN_THREADS")) nt = (unsigned)atoi(e);
    nt = std::max(1u, std::min(nt, 64u));
    if (nt == 1 || n < 64) {
        f(0, n);
        return;
    }
    std::vector<std::thread> th;
    const int64_t per = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        const int64_t lo = t * per, hi = std::min<int64_t>(n, lo + per);
        if (lo < hi) th.emplace_back([&f, lo, hi] { f(lo, hi); });
    }
    for (auto &t : th) t.join();
}

// f(i) for i in [0, n), one thread per index (n is small: the bands of a sweep plan)
template <class F>
void parallel_each(int64_t n, F &&f) {
    unsigned nt = std::thread::hardware_concurrency();
    if (const char *e = getenv("LGCN_PLAN_THREADS")) nt = (unsigned)atoi(e);
    if (nt <= 1 || n <= 1) {
        for (int64_t i = 0; i < n; ++i) f(i);
        return;
    }
    std::vector<std::thread> th;
    for (int64_t i = 0; i < n; ++i) th.emplace_back([&f, i] { f(i); });
    for (auto &t : th) t.join();
}

// Phases "pieces" and C of the sweep planner, from the run length of every (row, band) pair alone -- shared by the host
// planner (sweep_plan_build) and the device planner (lgc_sweep_dplan_*): piece cap, rounds, the piece list (slots
// contiguous per row, band-major), the rows' slot ranges, and the deal of the pieces over the wavefronts.
struct PlanPieces {
    int PCAP = 0, rounds = 0;
    int64_t n_waves = 0;
    std::vector<SweepPiece> pieces;
    std::vector<lgc_multi_row> multi;
    std::vector<std::vector<int32_t>> wave_pieces;
};

int plan_pieces_and_deal(const std::vector<int32_t> &run_len, int64_t n_rows, int32_t row_begin, const lgc_sweep_cfg &cfg,
                         PlanPieces &out) {
    const int NB = cfg.n_bands, WPBR = cfg.waves_per_band_round, CAP = cfg.row_cap;
    // piece length cap: the configured one, raised in steps of 16 while a longer cap saves a whole round
    auto rounds_for = [&](int64_t pcap) {
        std::vector<int64_t> per_band((size_t)NB, 0);
        for (int64_t i = 0; i < n_rows; ++i)
            for (int b = 0; b < NB; ++b) per_band[(size_t)b] += (run_len[(size_t)i * NB + b] + pcap - 1) / pcap;
        const int64_t most = *std::max_element(per_band.begin(), per_band.end());
        return (int)std::max<int64_t>(1, (most + (int64_t)WPBR * CAP - 1) / ((int64_t)WPBR * CAP));
    };
    int PCAP = cfg.piece_cap;
    const int fewest = rounds_for(int64_t(1) << 40);
    while (PCAP < 4 * cfg.piece_cap && rounds_for(PCAP) > fewest) PCAP += 16;
    const int rounds = rounds_for(PCAP);
    // pieces: long runs cut into near-equal parts of <= PCAP entries; slots are contiguous per row, band-major
    std::vector<SweepPiece> &pieces = out.pieces;
    int64_t row_start = 0;
    pieces.reserve((size_t)n_rows * NB + 1024);
    out.multi.resize((size_t)n_rows);
    for (int64_t i = 0; i < n_rows; ++i) {
        lgc_multi_row &mr = out.multi[(size_t)i];
        mr.row = row_begin + (int32_t)i;
        mr.slot_begin = (int32_t)pieces.size();
        mr.reserved = 0;
        int64_t k = row_start;                      // first entry of the row in the column-sorted array
        for (int b = 0; b < NB; ++b) {
            const int64_t cnt = run_len[(size_t)i * NB + b];
            if (cnt > 0) {
                const int64_t parts = (cnt + PCAP - 1) / PCAP, per = (cnt + parts - 1) / parts;
                for (int64_t q = 0; q < cnt; q += per)
                    pieces.push_back({(int32_t)(k + q), (int32_t)std::min<int64_t>(per, cnt - q), b});
            }
            k += cnt;
        }
        row_start = k;
        mr.slot_end = (int32_t)pieces.size();
    }
    const int64_t n_pieces = (int64_t)pieces.size();
    // C. pieces -> waves, band by band: heaviest first, dealt out in serpentine order (equal piece counts, close loads)
    std::vector<std::vector<int32_t>> by_band((size_t)NB);
    for (int64_t i = 0; i < n_pieces; ++i) by_band[(size_t)pieces[(size_t)i].band].push_back((int32_t)i);
    const int64_t U = (int64_t)rounds * WPBR;           // waves per band
    const int64_t n_waves = U * NB;
    std::vector<std::vector<int32_t>> &wave_pieces = out.wave_pieces;
    wave_pieces.assign((size_t)n_waves, {});
    // a band's pieces only ever go to that band's wavefronts (w encodes b): the bands are dealt out in parallel
    parallel_each(NB, [&](int64_t b) {
        auto &v = by_band[(size_t)b];
        std::stable_sort(v.begin(), v.end(), [&](int32_t a, int32_t c) { return pieces[(size_t)a].count > pieces[(size_t)c].count; });
        const bool by_weight = cfg.round_order >= 1;
        const size_t per_round = ((v.size() + (size_t)rounds - 1) / (size_t)rounds);
        for (size_t k = 0; k < v.size(); ++k) {
            int64_t u;
            if (by_weight) {   // heaviest pieces fill round 0, the next round 1, ...: serpentine inside the round
                const int64_t rr = (int64_t)(k / per_round), kk = (int64_t)(k % per_round);
                const int64_t lap = kk / WPBR, pos = kk % WPBR;
                u = rr * WPBR + ((lap & 1) ? (WPBR - 1 - pos) : pos);
            } else {
                const int64_t lap = (int64_t)(k / (size_t)U), pos = (int64_t)(k % (size_t)U);
                u = (lap & 1) ? (U - 1 - pos) : pos;
            }
            const int64_t r = u / WPBR, j = u % WPBR;
            // bands side by side: block = (r, j / 4, band), band = block % NB (one band per XCD, all bands at once)
            const int64_t w = ((r * (WPBR / 4) + j / 4) * NB + b) * 4 + (j % 4);
            wave_pieces[(size_t)w].push_back(v[k]);
        }
    });
    for (auto &wp : wave_pieces)
        if ((int)wp.size() > CAP) return LGC_E_INVAL;   // cannot happen: ceil(n / U) <= CAP
    out.PCAP = PCAP;
    out.rounds = rounds;
    out.n_waves = n_waves;
    return 0;
}

int sweep_plan_build(lgc_sweep_plan &pl, const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end,
                     int32_t col_lo, int32_t col_hi, const lgc_sweep_cfg &cfg) {
    const int NB = cfg.n_bands, WPBR = cfg.waves_per_band_round, CAP = cfg.row_cap;
    const bool timing = getenv("LGCN_PLAN_TIMING") != nullptr && atoi(getenv("LGCN_PLAN_TIMING")) != 0;
    auto clock0 = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[plan]   %-28s %.3f s\n", what, std::chrono::duration<double>(now - clock0).count());
        clock0 = now;
    };
    const int GROUPS = cfg.groups == 2 ? 2 : 4;          // entries per step = rows a wavefront gathers per instruction
    const int SLAB = 64 * GROUPS;                        // dwords per 32-step slab: 1 KiB (4 groups) or 512 B (2)
    const int64_t e0 = rowptr[row_begin], e1 = rowptr[row_end];
    const int64_t ne = e1 - e0;
    const int32_t n_rows = row_end - row_begin;
    // A. bands: column ranges with (nearly) equal entry counts
    std::vector<int32_t> hist((size_t)(col_hi - col_lo) + 1, 0);
    for (int64_t k = e0; k < e1; ++k) {
        const int32_t c = entries[k].col;
        if (c < col_lo || c >= col_hi) return LGC_E_INVAL;
        ++hist[c - col_lo];
    }
    std::vector<int32_t> bound(NB + 1, col_hi);   // band b = columns [bound[b], bound[b+1])
    bound[0] = col_lo;
    {
        int64_t run = 0;
        int b = 1;
        for (int32_t c = 0; c < col_hi - col_lo && b < NB; ++c) {
            run += hist[c];
            while (b < NB && run * NB >= ne * b && ne > 0) bound[b++] = col_lo + c + 1;
        }
    }
    lap("A bands (histogram)");
    // B. every row sorted by column (stable: equal columns keep edge order); run length of each (row, band)
    std::vector<lgc_entry> sorted((size_t)ne);
    std::vector<int32_t> run_len((size_t)n_rows * NB, 0);
    parallel_for(n_rows, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            const int64_t s = rowptr[row_begin + i] - e0, e = rowptr[row_begin + i + 1] - e0;
            std::copy(entries + e0 + s, entries + e0 + e, sorted.begin() + s);
            std::stable_sort(sorted.begin() + s, sorted.begin() + e,
                             [](const lgc_entry &a, const lgc_entry &b) { return a.col < b.col; });
            int64_t k = s;
            for (int b = 0; b < NB && k < e; ++b) {
                int64_t k2 = k;
                while (k2 < e && sorted[(size_t)k2].col < bound[b + 1]) ++k2;
                run_len[(size_t)i * NB + b] = (int32_t)(k2 - k);
                k = k2;
            }
        }
    });
    lap("B rows sorted by column");
    PlanPieces pp;
    {
        const int rc = plan_pieces_and_deal(run_len, n_rows, row_begin, cfg, pp);
        if (rc != 0) return rc;
    }
    const int PCAP = pp.PCAP, rounds = pp.rounds;
    std::vector<SweepPiece> &pieces = pp.pieces;
    pl.multi = pp.multi;
    const int64_t n_pieces = (int64_t)pieces.size(), n_waves = pp.n_waves;
    std::vector<std::vector<int32_t>> &wave_pieces = pp.wave_pieces;
    lap("C pieces -> waves");
    // D. per wave: merged column-sorted list -> conflict-free steps of 4 -> slabs of 32 steps
    const uint32_t PAD_X = 0x00FFFFFFu | ((uint32_t)CAP << 24);    // out-of-range column, dummy accumulator row
    std::vector<std::vector<uint32_t>> wave_slabs((size_t)n_waves);
    std::vector<int64_t> wave_steps((size_t)n_waves, 0), wave_pad((size_t)n_waves, 0);
    pl.wave_npieces.assign((size_t)n_waves, 0);
    pl.piece_slot.assign((size_t)n_waves * CAP, 0);
    const bool serpentine = cfg.round_order == 2;
    parallel_for(n_waves, [&](int64_t wlo, int64_t whi) {
        struct Item { int32_t col; float val; int32_t piece; };
        std::vector<Item> items;
        std::vector<char> done;
        for (int64_t w = wlo; w < whi; ++w) {
            auto &wp = wave_pieces[(size_t)w];
            auto &out = wave_slabs[(size_t)w];
            pl.wave_npieces[(size_t)w] = (int32_t)wp.size();
            items.clear();
            for (size_t lp = 0; lp < wp.size(); ++lp) {
                const SweepPiece &pc = pieces[(size_t)wp[lp]];
                pl.piece_slot[(size_t)w * CAP + lp] = wp[lp];
                for (int32_t k = 0; k < pc.count; ++k)
                    items.push_back({sorted[(size_t)(pc.begin + k)].col, sorted[(size_t)(pc.begin + k)].val, (int32_t)lp});
            }
            std::stable_sort(items.begin(), items.end(), [](const Item &a, const Item &b) { return a.col < b.col; });
            // odd rounds walk their band back down: what the previous round touched last is touched first, while it
            // is still in the Infinity Cache (and, for a few microseconds, in L2)
            if (serpentine && (((w / 4) / NB) / (WPBR / 4)) % 2 == 1) std::reverse(items.begin(), items.end());
            done.assign(items.size(), 0);
            size_t head = 0;
            int64_t step = 0, pad = 0;
            while (true) {
                while (head < items.size() && done[head]) ++head;
                if (head >= items.size()) break;
                if (step % 32 == 0) {
                    const size_t base = out.size();
                    out.resize(base + SLAB);
                    for (int i = 0; i < SLAB / 2; ++i) { out[base + 2 * i] = PAD_X; out[base + 2 * i + 1] = 0u; }
                }
                int used[4], n_used = 0, seen = 0;
                for (size_t i = head; n_used < GROUPS && i < items.size() && seen < cfg.lookahead; ++i) {
                    if (done[i]) continue;
                    ++seen;
                    bool clash = false;
                    for (int q = 0; q < n_used; ++q) clash |= used[q] == items[i].piece;
                    if (clash) continue;
                    const int grp = n_used;
                    used[n_used++] = items[i].piece;
                    done[i] = 1;
                    const int sl = (int)(step % 32);
                    // lane 16 g + (s >> 1), component pair (s & 1): dword (16 g + (s >> 1)) * 4 + 2 (s & 1) of the slab
                    const size_t at = (out.size() - SLAB) + (size_t)((16 * grp + (sl >> 1)) * 4 + 2 * (sl & 1));
                    uint32_t vbits;
                    memcpy(&vbits, &items[i].val, 4);
                    out[at] = ((uint32_t)items[i].col & 0xFFFFFFu) | ((uint32_t)items[i].piece << 24);
                    out[at + 1] = vbits;
                }
                pad += GROUPS - n_used;
                ++step;
            }
            wave_steps[(size_t)w] = step;
            wave_pad[(size_t)w] = pad;
        }
    });
    lap("D steps and slabs per wave");
    pl.wave_slab_ptr.assign((size_t)n_waves + 1, 0);
    int64_t total = 0, n_steps_total = 0, n_pad = 0;
    for (int64_t w = 0; w < n_waves; ++w) {
        pl.wave_slab_ptr[(size_t)w] = (int32_t)(total / SLAB);
        total += (int64_t)wave_slabs[(size_t)w].size();
        n_steps_total += wave_steps[(size_t)w];
        n_pad += wave_pad[(size_t)w];
    }
    if (total / SLAB >= INT32_MAX) return LGC_E_RANGE;
    pl.wave_slab_ptr[(size_t)n_waves] = (int32_t)(total / SLAB);
    pl.slabs.resize((size_t)total);                 // (zero-fill of ~85 MB: ~10 ms; the copies below overwrite all of it)
    parallel_for(n_waves, [&](int64_t wlo, int64_t whi) {
        for (int64_t w = wlo; w < whi; ++w)
            std::copy(wave_slabs[(size_t)w].begin(), wave_slabs[(size_t)w].end(),
                      pl.slabs.begin() + (size_t)pl.wave_slab_ptr[(size_t)w] * SLAB);
    });
    lap("slab concatenation");
    pl.dims.n_bands = NB;
    pl.dims.rounds = rounds;
    pl.dims.row_cap = CAP;
    pl.dims.piece_cap = PCAP;
    pl.dims.n_waves = n_waves;
    pl.dims.n_slabs = total / SLAB;
    pl.dims.groups = GROUPS;
    pl.dims.n_slots = n_pieces;
    pl.dims.n_rows = n_rows;
    pl.dims.n_entries = ne;
    pl.dims.n_steps = n_steps_total;
    pl.dims.n_padding = n_pad;
    return 0;
}

// Tuning switches for A/B runs, read ONCE per process (the launch paths used to call getenv per hop).
struct Knobs {
    bool no_fast_tiles, no_fused_apply;
    int64_t sweep_launch_waves;
};
const Knobs &knobs() {
    static const Knobs k = [] {
        Knobs v{};
        v.no_fast_tiles = getenv("LGCN_NO_FAST_TILES") != nullptr;
        v.no_fused_apply = getenv("LGCN_NO_FUSED_APPLY") != nullptr;
        const char *r = getenv("LGCN_SWEEP_LAUNCH_WAVES");
        v.sweep_launch_waves = r ? atoll(r) : 0;
        return v;
    }();
    return k;
}

// The opt-in for more than 64 KiB of dynamic LDS is per DEVICE: remember it per device ordinal.
int allow_big_lds(const void *fn, int bytes, unsigned long long *done_mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev >= 0 && ((*done_mask >> dev) & 1ull)) return 0;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0) *done_mask |= 1ull << dev;
    return 0;
}

// Arguments of one tile class + whether the DPP/buffer fast path applies (see lgc_spmm_tiles).
struct TilePrep {
    TileArgs p;
    bool fast;
    int64_t blocks;
};

int prepare_tiles(TilePrep &out, const int32_t *order, const int32_t *meta, const lgc_entry *slab, int32_t n_tiles, int32_t width,
                  int32_t tiles_per_wave, int64_t table_rows, const float *x, int64_t x_stride, float *y,
                  int64_t y_stride, const float *r, int64_t r_stride, float a, float b, int32_t dim) {
    if (dim < 4 || dim > 256) return LGC_E_DIM;
    if (!order || !slab || !x || !y || n_tiles < 0 || tiles_per_wave < 1 || x == y) return LGC_E_INVAL;
    if (width != 8 && width != 16 && width != 32) return LGC_E_INVAL;
    if (x_stride < dim || y_stride < dim || (r && r_stride < dim)) return LGC_E_INVAL;
    if (!aligned_to(x, 4) || !aligned_to(y, 4) || (r && !aligned_to(r, 4)) || !aligned_to(slab, 16)) return LGC_E_ALIGN;
    TileArgs p{};
    p.order = order;
    p.slab = reinterpret_cast<const u4 *>(slab);
    p.x = x; p.y = y; p.r = r;
    p.x_stride = x_stride; p.y_stride = y_stride; p.r_stride = r_stride;
    p.a = a; p.b = b;
    p.n_tiles = n_tiles; p.tiles_per_wave = tiles_per_wave;
    p.dim = dim;
    p.lpr = (dim + 3) / 4;
    p.wt_store = (table_rows * y_stride * 4 < (int64_t(1) << 32)) ? 1 : 0;
    p.meta = meta;
    const int64_t waves = ((int64_t)n_tiles + tiles_per_wave - 1) / tiles_per_wave;
    out.blocks = (waves + 3) / 4;
    // fast path: 16-lane rows, 24-bit row ids, 32-bit byte offsets, and the padding id 0xFFFFFF must fall outside
    // every table so that the hardware's range check turns padding into "load zeros / drop the store"
    auto table_bytes = [&](int64_t stride) { return ((table_rows - 1) * stride + dim) * 4; };
    auto pad_is_oob = [&](int64_t stride) {
        const uint32_t pad = (uint32_t)(0xFFFFFFull * (uint64_t)(stride * 4));
        return stride * 4 < (1 << 24) && table_bytes(stride) < (int64_t(1) << 32) && (int64_t)pad >= table_bytes(stride);
    };
    out.fast = meta != nullptr && ((dim >= 61 && dim <= 64) || (dim >= 68 && dim <= 128)) && table_rows > 0 &&
               table_rows < 0xFFFFFF &&
               pad_is_oob(x_stride) && pad_is_oob(y_stride) && (!r || pad_is_oob(r_stride)) && !knobs().no_fast_tiles;
    if (out.fast) {
        p.x_bytes = (uint32_t)table_bytes(x_stride);
        p.y_bytes = (uint32_t)table_bytes(y_stride);
        p.r_bytes = r ? (uint32_t)table_bytes(r_stride) : 0u;
    }
    out.p = p;
    return 0;
}


// ----------------------------------------------------------------------------------------
// The sweep planner with its bulk work on the device (lgc_sweep_dplan_*): same plan, bit for bit, as sweep_plan_build
// ----------------------------------------------------------------------------------------
// The host planner copies the half's 81 MB of entries to the host, sorts and scans them there and copies 87 MB of slabs
// back: 0.11-0.15 s.  Here the entries never leave the device:
//   A  column histogram (k_dp_hist) -> host: band bounds (a 1.6 M-element prefix walk)
//   B  every row sorted by column = ONE stable radix sort of (row << 24 | col) keys; run length of every (row, band)
//   -> host: pieces, rounds, deal (plan_pieces_and_deal: needs the run lengths only, 1.7 MB)
//   D  every wavefront's merged column-sorted list = ONE stable radix sort of (wave << 40 | col << 16 | piece << 8 | position
//      in the piece) keys (the low 40 bits inverted for wavefronts that walk their band downwards); then the greedy
//      conflict-free step builder, one WAVEFRONT per list: the host's scan looks at the next `lookahead` <= 64 undone
//      entries -- exactly one wavefront of lanes.  A step takes the first undone entry, then the first whose piece differs,
//      ... (<= GROUPS ballots), drops the taken lanes, closes the gaps through LDS and refills from the list.  Two passes:
//      count the steps of every list (-> slab offsets), then write the slabs.
__global__ void k_dp_hist(const lgc_entry *__restrict__ entries, int64_t e0, int64_t ne, int32_t col_lo, int32_t n_cols,
                          int32_t *__restrict__ hist, int32_t *__restrict__ bad) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ne) return;
    const int32_t c = entries[e0 + k].col - col_lo;
    if (c < 0 || c >= n_cols) *bad = 1;
    else atomicAdd(&hist[c], 1);
}

__global__ void k_dp_row_keys(const int32_t *__restrict__ rowptr, int32_t row_begin, int32_t n_rows,
                              const lgc_entry *__restrict__ entries, int64_t e0, int64_t ne,
                              unsigned long long *__restrict__ keys, int32_t *__restrict__ idx) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ne) return;
    int lo = 0, hi = n_rows - 1;                      // the row whose entry range holds e0 + k
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int64_t)rowptr[row_begin + mid] - e0 <= k) lo = mid;
        else hi = mid - 1;
    }
    keys[k] = ((unsigned long long)lo << 24) | (unsigned)(entries[e0 + k].col & 0xFFFFFF);
    idx[k] = (int32_t)k;
}

struct DpBounds { int32_t b[65]; };

__global__ void k_dp_gather(const lgc_entry *__restrict__ entries, int64_t e0, int64_t ne,
                            const unsigned long long *__restrict__ keys_sorted, const int32_t *__restrict__ idx_sorted,
                            DpBounds bound, int32_t nb, lgc_entry *__restrict__ sorted, int32_t *__restrict__ run_len) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ne) return;
    const lgc_entry ent = entries[e0 + idx_sorted[k]];
    sorted[k] = ent;
    const int64_t row = (int64_t)(keys_sorted[k] >> 24);
    int b = 0;
    while (b + 1 < nb && ent.col >= bound.b[b + 1]) ++b;
    atomicAdd(&run_len[row * nb + b], 1);
}

struct DpPiece { int32_t begin, count, wave, lp; };

__global__ void k_dp_item_keys(const DpPiece *__restrict__ pieces, int64_t n_pieces, const lgc_entry *__restrict__ sorted,
                               int32_t wpbr, int32_t nb, int32_t serpentine, unsigned long long *__restrict__ keys,
                               int32_t *__restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pieces) return;
    const DpPiece pc = pieces[i];
    const bool down = serpentine && (((pc.wave / 4) / nb) / (wpbr / 4)) % 2 == 1;
    const unsigned long long mask = (1ull << 40) - 1;
    for (int32_t j = 0; j < pc.count; ++j) {
        unsigned long long low = ((unsigned long long)(unsigned)(sorted[pc.begin + j].col & 0xFFFFFF) << 16) |
                                 ((unsigned long long)pc.lp << 8) | (unsigned)j;
        if (down) low = ~low & mask;
        keys[pc.begin + j] = ((unsigned long long)pc.wave << 40) | low;
        idx[pc.begin + j] = pc.begin + j;
    }
}

template <bool WRITE>
__global__ __launch_bounds__(kBlock) void k_dp_steps(const unsigned long long *__restrict__ keys, const int32_t *__restrict__ idx,
                                                    const lgc_entry *__restrict__ sorted, const int32_t *__restrict__ wave_item_ptr,
                                                    int32_t n_waves, int32_t groups, int32_t lookahead, int32_t wpbr, int32_t nb,
                                                    int32_t serpentine, uint32_t pad_x, int32_t *__restrict__ wave_steps,
                                                    int32_t *__restrict__ wave_pad, const int32_t *__restrict__ wave_slab_ptr,
                                                    uint32_t *__restrict__ slabs) {
    __shared__ uint32_t buf_cp[kBlock], buf_v[kBlock];
    const int lane = threadIdx.x & (kWave - 1), wib = threadIdx.x / kWave;
    const int w = blockIdx.x * (kBlock / kWave) + wib;
    if (w >= n_waves) return;                                     // wave-uniform
    const int32_t begin = wave_item_ptr[w], end = wave_item_ptr[w + 1];
    const bool down = serpentine && (((w / 4) / nb) / (wpbr / 4)) % 2 == 1;
    const unsigned long long mask = (1ull << 40) - 1;
    const int slab_dwords = 64 * groups;
    uint32_t *cp_buf = buf_cp + wib * kWave, *v_buf = buf_v + wib * kWave;
    auto load = [&](int32_t t, uint32_t &cp, uint32_t &v) {       // list item t: column | piece << 24, value bits
        unsigned long long low = keys[t] & mask;
        if (down) low = ~low & mask;
        cp = (uint32_t)((low >> 16) & 0xFFFFFF) | ((uint32_t)((low >> 8) & 0xFF) << 24);
        v = __float_as_uint(sorted[idx[t]].val);
    };
    int32_t next = begin;
    uint32_t cp = 0, v = 0;
    bool valid = lane < lookahead && next + lane < end;
    if (valid) load(next + lane, cp, v);
    next += min(lookahead, end - next);
    int32_t step = 0, pad = 0;
    while (true) {
        const unsigned long long live = __ballot(valid);
        if (live == 0) break;
        const int sl = step & 31;
        uint32_t *slab = nullptr;
        if constexpr (WRITE) {
            slab = slabs + ((int64_t)wave_slab_ptr[w] + (step >> 5)) * slab_dwords;
            if (sl == 0) {                                        // a fresh slab: every step padded until overwritten
                for (int i = lane; i < slab_dwords / 2; i += kWave) { slab[2 * i] = pad_x; slab[2 * i + 1] = 0u; }
                __builtin_amdgcn_s_waitcnt(0);                    // the padding lands before this wavefront's entries
            }
        }
        unsigned long long remaining = live, taken = 0;
        int n_used = 0;
        for (int g = 0; g < groups && remaining != 0; ++g) {
            const int j = __builtin_ctzll(remaining);             // the first undone entry whose piece is still free
            const uint32_t pj = __shfl(cp, j) >> 24;
            taken |= 1ull << j;
            if constexpr (WRITE) {
                if (lane == j) {
                    const int at = (16 * g + (sl >> 1)) * 4 + 2 * (sl & 1);
                    slab[at] = cp;
                    slab[at + 1] = v;
                }
            }
            remaining &= ~__ballot(valid && (cp >> 24) == pj);
            ++n_used;
        }
        pad += groups - n_used;
        ++step;
        // close the gaps: survivors keep their order, the list refills the tail of the window
        const unsigned long long surv = live & ~taken;
        const int n_surv = __builtin_popcountll(surv);
        const bool keep = (surv >> lane) & 1;
        if (keep) {
            const int pos = __builtin_popcountll(surv & ((1ull << lane) - 1));
            cp_buf[pos] = cp;
            v_buf[pos] = v;
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if (lane < n_surv) {
            cp = cp_buf[lane];
            v = v_buf[lane];
            valid = true;
        } else {
            const int32_t t = next + (lane - n_surv);
            valid = lane < lookahead && t < end;
            if (valid) load(t, cp, v);
        }
        __builtin_amdgcn_wave_barrier();
        next += min(max(lookahead - n_surv, 0), end - next);
    }
    if constexpr (!WRITE) {
        if (lane == 0) {
            wave_steps[w] = step;
            wave_pad[w] = pad;
        }
    }
}

}  // namespace

// ========================================================================================
// C ABI
// ========================================================================================
extern "C" {

int lgc_abi_version(void) { return LGC_ABI_VERSION; }

const char *lgc_error_string(int code) {
    switch (code) {
        case 0: return "ok";
        case LGC_E_INVAL: return "invalid argument";
        case LGC_E_DIM: return "unsupported embedding width";
        case LGC_E_WORKSPACE: return "workspace too small";
        case LGC_E_RANGE: return "node or edge count does not fit int32";
        case LGC_E_ALIGN: return "pointer or stride alignment violated";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int lgc_dim_ok(int32_t dim) {
    DimCfg cfg;
    return dim_cfg(dim, &cfg) ? 1 : 0;
}

size_t lgc_build_workspace_bytes(int64_t n_nodes, int64_t n_edges) {
    if (n_nodes < 0 || n_edges < 0 || n_nodes >= INT32_MAX || n_edges >= INT32_MAX) return 0;
    return carve_build_ws(nullptr, n_nodes, n_edges > 0 ? n_edges : 1).total;
}

int lgc_build_csr(const int64_t *edge_index, const float *edge_weight, int64_t n_nodes, int64_t n_edges,
                  int32_t by_source, int32_t normalize, const float *dis_in, int32_t *rowptr,
                  lgc_entry *entries, float *edge_val, float *deg_out, float *dis_out, void *workspace,
                  size_t workspace_bytes, int32_t *status, void *stream_) {
    if (n_nodes < 0 || n_edges < 0) return LGC_E_INVAL;
    if (n_nodes >= INT32_MAX || n_edges >= INT32_MAX) return LGC_E_RANGE;
    if (!rowptr || !status || (n_edges > 0 && (!edge_index || !entries || !workspace))) return LGC_E_INVAL;
    if (normalize && !dis_in && (by_source || !dis_out || !deg_out)) return LGC_E_INVAL;
    hipStream_t stream = as_stream(stream_);
    const int32_t N = (int32_t)n_nodes, E = (int32_t)n_edges;
    if (E == 0) {
        hipError_t err = hipMemsetAsync(rowptr, 0, sizeof(int32_t) * (size_t)(N + 1), stream);
        if (err == hipSuccess && normalize && !dis_in) {
            err = hipMemsetAsync(deg_out, 0, sizeof(float) * (size_t)N, stream);
            if (err == hipSuccess) err = hipMemsetAsync(dis_out, 0, sizeof(float) * (size_t)N, stream);
        }
        return (int)err;
    }
    BuildWs ws = carve_build_ws(workspace, n_nodes, n_edges);
    if (workspace_bytes < ws.total) return LGC_E_WORKSPACE;
    if (!aligned_to(workspace, 256)) return LGC_E_ALIGN;

    const int eb = ceil_div(E, kBlock), nb = ceil_div((long)N + 1, kBlock);
    hipLaunchKernelGGL(k_build_keys, dim3(eb), dim3(kBlock), 0, stream, edge_index, n_edges, n_nodes,
                       (int)by_source, ws.keys_in, ws.ids_in, status);
    size_t cub_bytes = ws.cub_bytes;
    hipError_t err = hipcub::DeviceRadixSort::SortPairs(ws.cub, cub_bytes, ws.keys_in, ws.keys_out, ws.ids_in,
                                                        ws.ids_out, E, 0, end_bit_for(n_nodes), stream);
    if (err != hipSuccess) return (int)err;
    hipLaunchKernelGGL(k_build_rowptr, dim3(nb), dim3(kBlock), 0, stream, ws.keys_out, E, N, rowptr);

    const float *dis = dis_in;
    if (normalize && !dis_in) {
        hipLaunchKernelGGL(k_build_sorted_weight, dim3(eb), dim3(kBlock), 0, stream, ws.ids_out, edge_weight, E,
                           ws.w_sorted);
        hipLaunchKernelGGL(k_build_degree_serial, dim3(ceil_div(N, kBlock)), dim3(kBlock), 0, stream, rowptr,
                           ws.w_sorted, N, deg_out);
        hipLaunchKernelGGL(k_build_degree_wave, dim3(ceil_div(N, kBlock / kWave)), dim3(kBlock), 0, stream, rowptr,
                           ws.w_sorted, N, deg_out);
        hipLaunchKernelGGL(k_build_dis, dim3(ceil_div(N, kBlock)), dim3(kBlock), 0, stream, deg_out, N, dis_out);
        dis = dis_out;
    }
    hipLaunchKernelGGL(k_build_entries, dim3(eb), dim3(kBlock), 0, stream, edge_index, edge_weight, ws.ids_out, dis,
                       n_edges, n_nodes, (int)by_source, (int)normalize, entries, edge_val);
    return (int)hipGetLastError();
}

int lgc_spmm(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end, int32_t short_max,
             const lgc_chunk *chunks, int32_t n_chunks, const lgc_multi_row *multi, int32_t n_multi, float *partials,
             int64_t table_rows, const float *x, int64_t x_stride, float *y, int64_t y_stride, const float *r, int64_t r_stride, float a,
             float b, int32_t dim, void *stream_) {
    DimCfg cfg;
    if (!dim_cfg(dim, &cfg)) return LGC_E_DIM;
    if (!rowptr || !x || !y || row_begin < 0 || row_end < row_begin || n_chunks < 0 || n_multi < 0 || short_max < 0 ||
        table_rows < row_end)
        return LGC_E_INVAL;
    if ((n_chunks > 0 && !chunks) || (n_multi > 0 && (!multi || !partials))) return LGC_E_INVAL;
    if (x_stride < dim || y_stride < dim || (r && r_stride < dim)) return LGC_E_INVAL;
    if (x == y) return LGC_E_INVAL;
    // dwordx4 accesses need only dword alignment on gfx950; 16-byte aligned rows (D % 4 == 0, torch
    // allocations) are the fast case, nothing else is rejected
    if (!aligned_to(x, 4) || !aligned_to(y, 4) || (r && !aligned_to(r, 4))) return LGC_E_ALIGN;
    hipStream_t stream = as_stream(stream_);
    SpmmArgs p{rowptr, entries, x, y, r, x_stride, y_stride, r_stride, a, b, dim, cfg.lpr, row_begin, row_end,
               short_max, 0};
    // write-through output stores address y with 32-bit byte offsets
    p.wt_store = (table_rows * y_stride * 4 < (int64_t(1) << 32)) ? 1 : 0;
    const int waves_per_block = kBlock / kWave;
    const int rows_per_wave = kWave / cfg.lpr;
    return dispatch_dim(cfg, [&](auto vec) -> int {
        constexpr int V = decltype(vec)::value;
        const int64_t n_rows = (int64_t)row_end - row_begin;
        const int row_blocks = n_rows > 0 ? ceil_div(ceil_div(n_rows, rows_per_wave), waves_per_block) : 0;
        const int chunk_blocks = n_chunks > 0 ? ceil_div(n_chunks, waves_per_block) : 0;
        const dim3 grid(row_blocks + chunk_blocks);
        if (grid.x > 0) {
            hipLaunchKernelGGL((k_spmm_hop<V>), grid, dim3(kBlock), 0, stream, p, chunks, n_chunks, partials, chunk_blocks);
        }
        if (n_multi > 0) {
            int blocks = ceil_div(n_multi, waves_per_block);
            hipLaunchKernelGGL((k_spmm_combine<V>), dim3(blocks), dim3(kBlock), 0, stream, p, multi, n_multi,
                               partials);
        }
        return (int)hipGetLastError();
    });
}

int lgc_spmm_rows(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end, const int64_t *row_ids,
                  int64_t n_ids, int64_t table_rows, const float *x, int64_t x_stride, float *y, int64_t y_stride, const float *r,
                  int64_t r_stride, float a, float b, int32_t dim, void *stream_) {
    DimCfg cfg;
    if (!dim_cfg(dim, &cfg)) return LGC_E_DIM;
    if (!rowptr || !x || !y || row_begin < 0 || row_end < row_begin || table_rows < row_end || n_ids < 0 || x == y) return LGC_E_INVAL;
    if (x_stride < dim || y_stride < dim || (r && r_stride < dim)) return LGC_E_INVAL;
    if (!aligned_to(x, 4) || !aligned_to(y, 4) || (r && !aligned_to(r, 4))) return LGC_E_ALIGN;
    if (n_ids == 0) return 0;
    if (!row_ids || !entries) return LGC_E_INVAL;
    SpmmArgs p{rowptr, entries, x, y, r, x_stride, y_stride, r_stride, a, b, dim, cfg.lpr, row_begin, row_end, 0, 0};
    return dispatch_dim(cfg, [&](auto vec) -> int {
        constexpr int V = decltype(vec)::value;
        hipLaunchKernelGGL((k_spmm_rows<V>), dim3(ceil_div(n_ids, kBlock / kWave)), dim3(kBlock), 0, as_stream(stream_), p, row_ids,
                           n_ids);
        return (int)hipGetLastError();
    });
}

int lgc_spmm_rows_split(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end, const int64_t *row_ids,
                        int64_t n_ids, int64_t table_rows, const float *x, int64_t x_stride, float *y, int64_t y_stride, int64_t y_rows,
                        const float *r, int64_t r_stride, float a, float b, int32_t dim, int32_t compact, int32_t *work,
                        float *partials, int64_t partial_rows, void *stream_) {
    DimCfg cfg;
    if (!dim_cfg(dim, &cfg)) return LGC_E_DIM;
    if (!rowptr || !x || !y || row_begin < 0 || row_end < row_begin || table_rows < row_end || n_ids < 0 || x == y) return LGC_E_INVAL;
    if (x_stride < dim || y_stride < dim || (r && r_stride < dim)) return LGC_E_INVAL;
    if (y_rows < (compact ? n_ids : (int64_t)row_end)) return LGC_E_INVAL;
    if (!aligned_to(x, 4) || !aligned_to(y, 4) || (r && !aligned_to(r, 4))) return LGC_E_ALIGN;
    if (n_ids == 0) return 0;
    if (!row_ids || !entries || !work || !partials) return LGC_E_INVAL;
    // every list position may end its row with a short chunk: at least one partial row per position and one to spare;
    // chunk numbers are int32
    if (partial_rows <= n_ids || partial_rows >= INT32_MAX || n_ids >= INT32_MAX - 2) return LGC_E_RANGE;
    if (!aligned_to(partials, 4)) return LGC_E_ALIGN;
    hipStream_t stream = as_stream(stream_);
    SpmmArgs p{rowptr, entries, x, y, r, x_stride, y_stride, r_stride, a, b, dim, cfg.lpr, row_begin, row_end, 0, 0};
    const int waves_per_block = kBlock / kWave;
    return dispatch_dim(cfg, [&](auto vec) -> int {
        constexpr int V = decltype(vec)::value;
        hipLaunchKernelGGL(k_rows_plan, dim3(1), dim3(kRowsPlanBlock), 0, stream, rowptr, row_ids, n_ids, row_begin, row_end,
                           partial_rows, work);
        // a fixed grid strides over the chunk numbers (their count is known on the device only): enough wavefronts for a
        // few thousand chunks to run side by side, every wavefront leaves the loop once the numbers run out
        const int64_t want = ceil_div(std::min<int64_t>(partial_rows, 8192), waves_per_block);
        hipLaunchKernelGGL((k_rows_chunks<V>), dim3((unsigned)std::max<int64_t>(want, 1)), dim3(kBlock), 0, stream, p, row_ids, n_ids,
                           work, partials, compact);
        hipLaunchKernelGGL((k_rows_combine<V>), dim3(ceil_div(n_ids, waves_per_block)), dim3(kBlock), 0, stream, p, row_ids, n_ids,
                           work, partials, compact);
        return (int)hipGetLastError();
    });
}

int lgc_bipartite_split(const int64_t *edge_index, int64_t n_edges, int64_t *out2, void *stream_) {
    if (!out2 || n_edges < 0 || (n_edges > 0 && !edge_index)) return LGC_E_INVAL;
    hipStream_t st = as_stream(stream_);
    hipLaunchKernelGGL(k_split_init, dim3(1), dim3(1), 0, st, reinterpret_cast<long long *>(out2));
    if (n_edges > 0)
        hipLaunchKernelGGL(k_bipartite_split, dim3((unsigned)std::min<int64_t>(ceil_div(n_edges, kBlock), 256 * 16)), dim3(kBlock), 0,
                           st, edge_index, n_edges, reinterpret_cast<long long *>(out2));
    return (int)hipGetLastError();
}

static size_t row_plan_ws(void *base, int64_t n, int32_t **a, int32_t **b, int32_t **c, void **cub, size_t *cub_bytes) {
    size_t cb = 0;
    int32_t *nul = nullptr;
    (void)hipcub::DeviceScan::ExclusiveSum(nullptr, cb, nul, nul, (int)std::max<int64_t>(n, 1), (hipStream_t)0);
    const uintptr_t p = reinterpret_cast<uintptr_t>(base);
    const size_t arr = align_up((size_t)(n + 1) * 4, 256);
    if (a) *a = reinterpret_cast<int32_t *>(p);
    if (b) *b = reinterpret_cast<int32_t *>(p + arr);
    if (c) *c = reinterpret_cast<int32_t *>(p + 2 * arr);
    if (cub) *cub = reinterpret_cast<void *>(p + 3 * arr);
    if (cub_bytes) *cub_bytes = cb;
    return 3 * arr + align_up(cb, 256) + 256;
}

size_t lgc_row_plan_workspace_bytes(int64_t n_rows) {
    if (n_rows < 0 || n_rows >= INT32_MAX) return 0;
    return row_plan_ws(nullptr, n_rows, nullptr, nullptr, nullptr, nullptr, nullptr);
}

// Phase 1: per-row counts and their exclusive prefix sums (one extra element each = the totals), left in the workspace;
// totals[3] (int32, device) = chunks, multi-chunk rows, slots.  Phase 2 (lgc_row_plan_fill) writes the two lists.
int lgc_row_plan_count(const int32_t *rowptr, int32_t row_begin, int32_t row_end, int32_t short_max, int32_t chunk_len,
                       void *workspace, size_t workspace_bytes, int32_t *totals, void *stream_) {
    if (!rowptr || !totals || row_begin < 0 || row_end < row_begin || short_max < 0 || chunk_len < 1) return LGC_E_INVAL;
    const int64_t n = (int64_t)row_end - row_begin;
    hipStream_t st = as_stream(stream_);
    if (n > 0) {
        if (!workspace) return LGC_E_INVAL;
        if (workspace_bytes < lgc_row_plan_workspace_bytes(n)) return LGC_E_WORKSPACE;
        if (!aligned_to(workspace, 256)) return LGC_E_ALIGN;
    }
    hipError_t err = hipMemsetAsync(totals, 0, 3 * sizeof(int32_t), st);
    if (err != hipSuccess) return (int)err;
    if (n == 0) return 0;
    int32_t *a, *b, *c;
    void *cub;
    size_t cub_bytes;
    row_plan_ws(workspace, n, &a, &b, &c, &cub, &cub_bytes);
    // one extra trailing zero per array, so that the exclusive sum's element n is the total
    for (int32_t *arr : {a, b, c}) {
        err = hipMemsetAsync(arr + n, 0, sizeof(int32_t), st);
        if (err != hipSuccess) return (int)err;
    }
    hipLaunchKernelGGL(k_row_plan_count, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, st, rowptr, row_begin, row_end, short_max,
                       chunk_len, a, b, c);
    int k = 0;
    for (int32_t *arr : {a, b, c}) {
        err = hipcub::DeviceScan::ExclusiveSum(cub, cub_bytes, arr, arr, (int)(n + 1), st);
        if (err == hipSuccess) err = hipMemcpyAsync(totals + k, arr + n, sizeof(int32_t), hipMemcpyDeviceToDevice, st);
        if (err != hipSuccess) return (int)err;
        ++k;
    }
    return (int)hipGetLastError();
}

int lgc_row_plan_fill(const int32_t *rowptr, int32_t row_begin, int32_t row_end, int32_t short_max, int32_t chunk_len,
                      const void *workspace, lgc_chunk *chunks, lgc_multi_row *multi, void *stream_) {
    if (!rowptr || row_begin < 0 || row_end < row_begin || short_max < 0 || chunk_len < 1) return LGC_E_INVAL;
    const int64_t n = (int64_t)row_end - row_begin;
    if (n == 0) return 0;
    if (!workspace || !chunks || !multi) return LGC_E_INVAL;
    int32_t *a, *b, *c;
    row_plan_ws(const_cast<void *>(workspace), n, &a, &b, &c, nullptr, nullptr);
    hipLaunchKernelGGL(k_row_plan_fill, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, as_stream(stream_), rowptr, row_begin, row_end,
                       short_max, chunk_len, a, b, c, chunks, multi);
    return (int)hipGetLastError();
}

static size_t tile_classes_ws(void *base, int64_t n_rows, int64_t table_rows, int32_t **pop, unsigned long long **k_in,
                              unsigned long long **k_out, int32_t **v_in, void **cub, size_t *cub_bytes) {
    size_t cb = 0;
    unsigned long long *nk = nullptr;
    int32_t *nv = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, cb, nk, nk, nv, nv, (int)std::max<int64_t>(n_rows, 1), 0, 60, (hipStream_t)0);
    const uintptr_t p = reinterpret_cast<uintptr_t>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += align_up(bytes, 256); return p + at; };
    const uintptr_t a_pop = take((size_t)table_rows * 4), a_ki = take((size_t)n_rows * 8), a_ko = take((size_t)n_rows * 8),
                    a_vi = take((size_t)n_rows * 4), a_cub = take(cb);
    if (pop) *pop = reinterpret_cast<int32_t *>(a_pop);
    if (k_in) *k_in = reinterpret_cast<unsigned long long *>(a_ki);
    if (k_out) *k_out = reinterpret_cast<unsigned long long *>(a_ko);
    if (v_in) *v_in = reinterpret_cast<int32_t *>(a_vi);
    if (cub) *cub = reinterpret_cast<void *>(a_cub);
    if (cub_bytes) *cub_bytes = cb;
    return off + 256;
}

size_t lgc_tile_classes_workspace_bytes(int64_t n_rows, int64_t table_rows) {
    if (n_rows < 0 || table_rows < 0 || n_rows >= INT32_MAX || table_rows >= INT32_MAX) return 0;
    return tile_classes_ws(nullptr, n_rows, table_rows, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
}

int lgc_tile_classes(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end, int32_t max_len,
                     int32_t cold, int64_t table_rows, void *workspace, size_t workspace_bytes, int32_t *sorted_rows,
                     uint64_t *class_count, void *stream_) {
    if (!rowptr || !sorted_rows || !class_count || row_begin < 0 || row_end < row_begin || table_rows < row_end || max_len < 0 ||
        table_rows >= INT32_MAX)
        return LGC_E_INVAL;
    const int64_t n = (int64_t)row_end - row_begin;
    hipStream_t st = as_stream(stream_);
    if (n > 0) {                                  // every argument is checked before the first call into the runtime
        if (!workspace) return LGC_E_INVAL;       // (entries may be NULL for rows without a single entry: checked below)
        if (workspace_bytes < lgc_tile_classes_workspace_bytes(n, table_rows)) return LGC_E_WORKSPACE;
        if (!aligned_to(workspace, 256)) return LGC_E_ALIGN;
    }
    hipError_t err = hipMemsetAsync(class_count, 0, 4 * sizeof(uint64_t), st);
    if (err != hipSuccess) return (int)err;
    if (n == 0) return 0;
    int32_t *pop, *v_in;
    unsigned long long *k_in, *k_out;
    void *cub;
    size_t cub_bytes;
    tile_classes_ws(workspace, n, table_rows, &pop, &k_in, &k_out, &v_in, &cub, &cub_bytes);
    const int32_t cap = std::min(max_len, 32);
    if (cold) {
        err = hipMemsetAsync(pop, 0, (size_t)table_rows * 4, st);
        if (err != hipSuccess) return (int)err;
        // the entry range of the rows is known on the device only: one thread per entry of the whole CSR at most is
        // wasteful, so the launch is sized by an upper bound the caller's CSR gives for free -- rows hold <= 2^31 entries;
        // the kernel bounds itself by rowptr[row_end]
        int32_t ends[2];
        err = hipMemcpyAsync(&ends[0], rowptr + row_begin, 4, hipMemcpyDeviceToHost, st);
        if (err == hipSuccess) err = hipMemcpyAsync(&ends[1], rowptr + row_end, 4, hipMemcpyDeviceToHost, st);
        if (err == hipSuccess) err = hipStreamSynchronize(st);
        if (err != hipSuccess) return (int)err;
        const int64_t n_ent = (int64_t)ends[1] - ends[0];
        if (n_ent > 0 && !entries) return LGC_E_INVAL;
        if (n_ent > 0)
            hipLaunchKernelGGL(k_tile_pop, dim3(ceil_div(n_ent, kBlock)), dim3(kBlock), 0, st, rowptr, entries, row_begin, row_end,
                               pop);
    }
    hipLaunchKernelGGL(k_tile_keys, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, st, rowptr, entries, row_begin, row_end, cap,
                       cold ? 1 : 0, pop, k_in, v_in, reinterpret_cast<unsigned long long *>(class_count));
    err = hipcub::DeviceRadixSort::SortPairs(cub, cub_bytes, k_in, k_out, v_in, sorted_rows, (int)n, 0, 60, st);
    if (err != hipSuccess) return (int)err;
    return (int)hipGetLastError();
}

int lgc_tile_pack(const int32_t *rowptr, const int32_t *sorted_rows, int64_t n_rows, int32_t width, int32_t *order, int32_t *meta,
                  void *stream_) {
    if (!rowptr || n_rows < 0 || (width != 8 && width != 16 && width != 32)) return LGC_E_INVAL;
    if (n_rows == 0) return 0;
    if (!sorted_rows || !order || !meta) return LGC_E_INVAL;
    const int R = 128 * (width == 32 ? 2 : 1) / width;
    const int64_t n_tiles = (n_rows + R - 1) / R;
    hipLaunchKernelGGL(k_tile_pack, dim3(ceil_div(n_tiles, kBlock)), dim3(kBlock), 0, as_stream(stream_), rowptr, sorted_rows, n_rows,
                       R, n_tiles, order, meta);
    return (int)hipGetLastError();
}

int lgc_build_tiles(const int32_t *rowptr, const lgc_entry *entries, const int32_t *order, int64_t n_slots, int32_t width,
                    lgc_entry *slab, void *stream_) {
    if (!rowptr || !order || !slab || n_slots < 0 || (width != 8 && width != 16 && width != 32)) return LGC_E_INVAL;
    const int L = width == 32 ? 2 : 1;
    if (n_slots % (128 * L / width) != 0) return LGC_E_INVAL;
    if (n_slots == 0) return 0;
    hipLaunchKernelGGL(k_build_tiles, dim3(ceil_div(n_slots * width, kBlock)), dim3(kBlock), 0, as_stream(stream_), rowptr,
                       entries, order, n_slots, width, L, slab);
    return (int)hipGetLastError();
}

int lgc_spmm_tiles(const int32_t *order, const int32_t *meta, const lgc_entry *slab, int32_t n_tiles, int32_t width,
                   int32_t tiles_per_wave, int64_t table_rows, const float *x, int64_t x_stride, float *y, int64_t y_stride,
                   const float *r, int64_t r_stride, float a, float b, int32_t dim, void *stream_) {
    TilePrep tp;
    const int rc = prepare_tiles(tp, order, meta, slab, n_tiles, width, tiles_per_wave, table_rows, x, x_stride, y, y_stride,
                                 r, r_stride, a, b, dim);
    if (rc != 0) return rc;
    if (n_tiles == 0) return 0;
    hipStream_t stream = as_stream(stream_);
    const dim3 grid((unsigned)tp.blocks);
    if (tp.fast) {
        if (width == 8) hipLaunchKernelGGL((k_rows_tile_dpp<8, 1>), grid, dim3(kBlock), 0, stream, tp.p);
        else if (width == 16) hipLaunchKernelGGL((k_rows_tile_dpp<16, 1>), grid, dim3(kBlock), 0, stream, tp.p);
        else hipLaunchKernelGGL((k_rows_tile_dpp<32, 2>), grid, dim3(kBlock), 0, stream, tp.p);
    } else {
        if (width == 8) hipLaunchKernelGGL((k_rows_tile<8, 1>), grid, dim3(kBlock), 0, stream, tp.p);
        else if (width == 16) hipLaunchKernelGGL((k_rows_tile<16, 1>), grid, dim3(kBlock), 0, stream, tp.p);
        else hipLaunchKernelGGL((k_rows_tile<32, 2>), grid, dim3(kBlock), 0, stream, tp.p);
    }
    return (int)hipGetLastError();
}

lgc_sweep_plan *lgc_sweep_plan_create(const int32_t *rowptr_host, const lgc_entry *entries_host, int32_t row_begin,
                                      int32_t row_end, int32_t col_lo, int32_t col_hi, const lgc_sweep_cfg *cfg, int *code) {
    int rc = LGC_E_INVAL;
    lgc_sweep_plan *pl = nullptr;
    if (rowptr_host && (entries_host || rowptr_host[row_end] == rowptr_host[row_begin]) && cfg && row_begin >= 0 &&
        row_end >= row_begin && col_lo >= 0 && col_hi > col_lo && col_hi <= 0xFFFFFF && cfg->n_bands >= 1 &&
        cfg->n_bands <= 64 && cfg->waves_per_band_round >= 4 && cfg->waves_per_band_round % 4 == 0 && cfg->row_cap >= 1 &&
        cfg->row_cap <= 254 && cfg->piece_cap >= 1 && cfg->lookahead >= 4 && (cfg->groups == 0 || cfg->groups == 2 || cfg->groups == 4) &&
        cfg->round_order >= 0 && cfg->round_order <= 2) {
        pl = new (std::nothrow) lgc_sweep_plan();
        if (pl) {
            try {
                rc = sweep_plan_build(*pl, rowptr_host, entries_host, row_begin, row_end, col_lo, col_hi, *cfg);
            } catch (const std::bad_alloc &) {
                rc = (int)hipErrorOutOfMemory;
            }
            if (rc != 0) { delete pl; pl = nullptr; }
        } else {
            rc = (int)hipErrorOutOfMemory;
        }
    }
    if (code) *code = rc;
    return pl;
}

int lgc_sweep_plan_dims(const lgc_sweep_plan *plan, lgc_sweep_dims *dims) {
    if (!plan || !dims) return LGC_E_INVAL;
    *dims = plan->dims;
    return 0;
}

int lgc_sweep_plan_export(const lgc_sweep_plan *plan, uint32_t *slabs, int32_t *wave_slab_ptr, int32_t *wave_npieces,
                          int32_t *piece_slot, lgc_multi_row *multi) {
    if (!plan || !slabs || !wave_slab_ptr || !wave_npieces || !piece_slot || !multi) return LGC_E_INVAL;
    std::copy(plan->slabs.begin(), plan->slabs.end(), slabs);
    std::copy(plan->wave_slab_ptr.begin(), plan->wave_slab_ptr.end(), wave_slab_ptr);
    std::copy(plan->wave_npieces.begin(), plan->wave_npieces.end(), wave_npieces);
    std::copy(plan->piece_slot.begin(), plan->piece_slot.end(), piece_slot);
    std::copy(plan->multi.begin(), plan->multi.end(), multi);
    return 0;
}

int lgc_sweep_plan_export_multi(const lgc_sweep_plan *plan, lgc_multi_row *multi) {
    if (!plan || !multi) return LGC_E_INVAL;
    std::copy(plan->multi.begin(), plan->multi.end(), multi);
    return 0;
}

int lgc_sweep_plan_upload(const lgc_sweep_plan *plan, uint32_t *slabs, int32_t *wave_slab_ptr, int32_t *wave_npieces,
                          int32_t *piece_slot, void *stream_) {
    if (!plan || !slabs || !wave_slab_ptr || !wave_npieces || !piece_slot) return LGC_E_INVAL;
    hipStream_t st = as_stream(stream_);
    auto up = [&](void *dst, const void *src, size_t bytes) -> int {
        return bytes == 0 ? 0 : (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    };
    int rc = up(slabs, plan->slabs.data(), plan->slabs.size() * sizeof(uint32_t));
    if (rc == 0) rc = up(wave_slab_ptr, plan->wave_slab_ptr.data(), plan->wave_slab_ptr.size() * sizeof(int32_t));
    if (rc == 0) rc = up(wave_npieces, plan->wave_npieces.data(), plan->wave_npieces.size() * sizeof(int32_t));
    if (rc == 0) rc = up(piece_slot, plan->piece_slot.data(), plan->piece_slot.size() * sizeof(int32_t));
    if (rc != 0) return rc;
    return (int)hipStreamSynchronize(st);       // the plan's host arrays may be freed as soon as this returns
}

void lgc_sweep_plan_free(lgc_sweep_plan *plan) { delete plan; }


// ---- device planner (see the kernels k_dp_*) ----------------------------------------------------------------------
namespace {
struct DplanWs {
    int32_t *hist, *run_len, *idx_a, *idx_b, *item_ptr, *steps, *pad, *slab_ptr, *bad;
    unsigned long long *keys_a, *keys_b;
    lgc_entry *sorted;
    DpPiece *pieces;
    void *cub;
    size_t cub_bytes, total;
};

int64_t dplan_max_pieces(int64_t ne, int64_t n_rows, const lgc_sweep_cfg &cfg) {
    return n_rows * cfg.n_bands + ne / std::max(1, cfg.piece_cap) + 16;
}

int64_t dplan_max_waves(int64_t ne, int64_t n_rows, const lgc_sweep_cfg &cfg) {
    const int64_t per_round = (int64_t)cfg.waves_per_band_round * cfg.row_cap;
    const int64_t rounds = (dplan_max_pieces(ne, n_rows, cfg) + per_round - 1) / per_round + 1;
    return rounds * cfg.waves_per_band_round * cfg.n_bands;
}

DplanWs dplan_carve(void *base, int64_t ne, int64_t n_rows, int64_t n_cols, const lgc_sweep_cfg &cfg) {
    DplanWs ws{};
    size_t cb = 0;
    unsigned long long *nk = nullptr;
    int32_t *nv = nullptr;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, cb, nk, nk, nv, nv, (int)std::max<int64_t>(ne, 1), 0, 64, (hipStream_t)0);
    const uintptr_t p = reinterpret_cast<uintptr_t>(base);
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t at = off; off += align_up(std::max<size_t>(bytes, 4), 256); return p + at; };
    const int64_t mw = dplan_max_waves(ne, n_rows, cfg);
    ws.hist = reinterpret_cast<int32_t *>(take((size_t)n_cols * 4));
    ws.run_len = reinterpret_cast<int32_t *>(take((size_t)n_rows * cfg.n_bands * 4));
    ws.keys_a = reinterpret_cast<unsigned long long *>(take((size_t)ne * 8));
    ws.keys_b = reinterpret_cast<unsigned long long *>(take((size_t)ne * 8));
    ws.idx_a = reinterpret_cast<int32_t *>(take((size_t)ne * 4));
    ws.idx_b = reinterpret_cast<int32_t *>(take((size_t)ne * 4));
    ws.sorted = reinterpret_cast<lgc_entry *>(take((size_t)ne * sizeof(lgc_entry)));
    ws.pieces = reinterpret_cast<DpPiece *>(take((size_t)dplan_max_pieces(ne, n_rows, cfg) * sizeof(DpPiece)));
    ws.item_ptr = reinterpret_cast<int32_t *>(take((size_t)(mw + 1) * 4));
    ws.steps = reinterpret_cast<int32_t *>(take((size_t)mw * 4));
    ws.pad = reinterpret_cast<int32_t *>(take((size_t)mw * 4));
    ws.slab_ptr = reinterpret_cast<int32_t *>(take((size_t)(mw + 1) * 4));
    ws.bad = reinterpret_cast<int32_t *>(take(4));
    ws.cub = reinterpret_cast<void *>(take(cb));
    ws.cub_bytes = cb;
    ws.total = off + 256;
    return ws;
}

bool dplan_cfg_ok(const lgc_sweep_cfg *cfg) {
    return cfg && cfg->n_bands >= 1 && cfg->n_bands <= 64 && cfg->waves_per_band_round >= 4 && cfg->waves_per_band_round % 4 == 0 &&
           cfg->row_cap >= 1 && cfg->row_cap <= 254 && cfg->piece_cap >= 1 && cfg->piece_cap <= 64 && cfg->lookahead >= 4 &&
           cfg->lookahead <= 64 && (cfg->groups == 0 || cfg->groups == 2 || cfg->groups == 4) && cfg->round_order >= 0 &&
           cfg->round_order <= 2;
}
}  // namespace

size_t lgc_sweep_dplan_workspace_bytes(int64_t n_entries, int64_t n_rows, int64_t n_cols, const lgc_sweep_cfg *cfg) {
    if (!dplan_cfg_ok(cfg) || n_entries < 0 || n_rows < 0 || n_cols <= 0 || n_entries >= INT32_MAX) return 0;
    return dplan_carve(nullptr, n_entries, n_rows, n_cols, *cfg).total;
}

lgc_sweep_dplan *lgc_sweep_dplan_create(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end,
                                        int64_t first_entry, int64_t n_entries, int32_t col_lo, int32_t col_hi,
                                        const lgc_sweep_cfg *cfg, void *workspace, size_t workspace_bytes, void *stream_,
                                        int *code) {
    auto fail = [&](int rc) -> lgc_sweep_dplan * {
        if (code) *code = rc;
        return nullptr;
    };
    // piece_cap <= 64 and lookahead <= 64: the composite sort key keeps 8 bits for an entry's position in its piece (pieces
    // hold up to 4 x piece_cap entries) and the step builder's window is one wavefront; other settings -> the host planner
    if (!dplan_cfg_ok(cfg)) return fail(LGC_E_RANGE);
    if (!rowptr || row_begin < 0 || row_end < row_begin || first_entry < 0 || n_entries < 0 || col_lo < 0 || col_hi <= col_lo ||
        col_hi > 0xFFFFFF || n_entries >= INT32_MAX || (n_entries > 0 && (!entries || !workspace)))
        return fail(LGC_E_INVAL);
    const int64_t ne = n_entries, e0 = first_entry, n_rows = (int64_t)row_end - row_begin, n_cols = (int64_t)col_hi - col_lo;
    const int NB = cfg->n_bands, WPBR = cfg->waves_per_band_round, CAP = cfg->row_cap, GROUPS = cfg->groups == 2 ? 2 : 4;
    const int SLAB = 64 * GROUPS;
    if (ne > 0) {
        if (workspace_bytes < lgc_sweep_dplan_workspace_bytes(ne, n_rows, n_cols, cfg)) return fail(LGC_E_WORKSPACE);
        if (!aligned_to(workspace, 256)) return fail(LGC_E_ALIGN);
    }
    hipStream_t st = as_stream(stream_);
    lgc_sweep_dplan *pl = new (std::nothrow) lgc_sweep_dplan();
    if (!pl) return fail((int)hipErrorOutOfMemory);
    auto bail = [&](int rc) -> lgc_sweep_dplan * {
        delete pl;
        return fail(rc);
    };
    try {
        pl->cfg = *cfg;
        pl->row_begin = row_begin;
        const DplanWs ws = dplan_carve(workspace, ne, n_rows, n_cols, *cfg);
        std::vector<int32_t> run_len((size_t)n_rows * NB, 0);
        if (ne > 0) {
            // A. histogram of the columns -> band bounds
            hipError_t err = hipMemsetAsync(ws.hist, 0, (size_t)n_cols * 4, st);
            if (err == hipSuccess) err = hipMemsetAsync(ws.run_len, 0, (size_t)n_rows * NB * 4, st);
            if (err == hipSuccess) err = hipMemsetAsync(ws.bad, 0, 4, st);
            if (err != hipSuccess) return bail((int)err);
            const int eb = ceil_div(ne, kBlock);
            hipLaunchKernelGGL(k_dp_hist, dim3(eb), dim3(kBlock), 0, st, entries, e0, ne, col_lo, (int32_t)n_cols, ws.hist, ws.bad);
            std::vector<int32_t> hist((size_t)n_cols);
            int32_t bad = 0;
            err = hipMemcpyAsync(hist.data(), ws.hist, (size_t)n_cols * 4, hipMemcpyDeviceToHost, st);
            if (err == hipSuccess) err = hipMemcpyAsync(&bad, ws.bad, 4, hipMemcpyDeviceToHost, st);
            // B meanwhile: rows sorted by column
            hipLaunchKernelGGL(k_dp_row_keys, dim3(eb), dim3(kBlock), 0, st, rowptr, row_begin, (int32_t)n_rows, entries, e0, ne,
                               ws.keys_a, ws.idx_a);
            size_t cb = ws.cub_bytes;
            int row_bits = 1;
            while ((int64_t(1) << row_bits) < n_rows) ++row_bits;
            if (err == hipSuccess)
                err = hipcub::DeviceRadixSort::SortPairs(ws.cub, cb, ws.keys_a, ws.keys_b, ws.idx_a, ws.idx_b, (int)ne, 0,
                                                         24 + row_bits, st);
            if (err == hipSuccess) err = hipStreamSynchronize(st);
            if (err != hipSuccess) return bail((int)err);
            if (bad) return bail(LGC_E_INVAL);                     // a column outside [col_lo, col_hi)
            DpBounds bound;
            for (int b = 0; b <= NB; ++b) bound.b[b] = col_hi;
            bound.b[0] = col_lo;
            {
                int64_t run = 0;
                int b = 1;
                for (int32_t c = 0; c < n_cols && b < NB; ++c) {
                    run += hist[(size_t)c];
                    while (b < NB && run * NB >= ne * b) bound.b[b++] = col_lo + c + 1;
                }
            }
            hipLaunchKernelGGL(k_dp_gather, dim3(eb), dim3(kBlock), 0, st, entries, e0, ne, ws.keys_b, ws.idx_b, bound, NB, ws.sorted,
                               ws.run_len);
            err = hipMemcpyAsync(run_len.data(), ws.run_len, run_len.size() * 4, hipMemcpyDeviceToHost, st);
            if (err == hipSuccess) err = hipStreamSynchronize(st);
            if (err != hipSuccess) return bail((int)err);
        }
        // pieces, rounds, deal: on the host, from the run lengths alone
        PlanPieces pp;
        const int rc = plan_pieces_and_deal(run_len, n_rows, row_begin, *cfg, pp);
        if (rc != 0) return bail(rc);
        if (pp.PCAP > 256) return bail(LGC_E_RANGE);
        const int64_t n_waves = pp.n_waves, n_pieces = (int64_t)pp.pieces.size();
        if (n_pieces > dplan_max_pieces(ne, n_rows, *cfg) || n_waves > dplan_max_waves(ne, n_rows, *cfg)) return bail(LGC_E_WORKSPACE);
        pl->multi = pp.multi;
        pl->wave_npieces.assign((size_t)n_waves, 0);
        pl->piece_slot.assign((size_t)n_waves * CAP, 0);
        pl->wave_item_ptr.assign((size_t)n_waves + 1, 0);
        std::vector<DpPiece> table((size_t)n_pieces);
        for (int64_t w = 0; w < n_waves; ++w) {
            const auto &wp = pp.wave_pieces[(size_t)w];
            pl->wave_npieces[(size_t)w] = (int32_t)wp.size();
            int32_t items = 0;
            for (size_t lp = 0; lp < wp.size(); ++lp) {
                const SweepPiece &pc = pp.pieces[(size_t)wp[lp]];
                pl->piece_slot[(size_t)w * CAP + lp] = wp[lp];
                table[(size_t)wp[lp]] = DpPiece{pc.begin, pc.count, (int32_t)w, (int32_t)lp};
                items += pc.count;
            }
            pl->wave_item_ptr[(size_t)w + 1] = pl->wave_item_ptr[(size_t)w] + items;
        }
        std::vector<int32_t> steps((size_t)n_waves, 0), pads((size_t)n_waves, 0);
        if (ne > 0 && n_waves > 0) {
            hipError_t err = hipMemcpyAsync(ws.pieces, table.data(), table.size() * sizeof(DpPiece), hipMemcpyHostToDevice, st);
            if (err == hipSuccess)
                err = hipMemcpyAsync(ws.item_ptr, pl->wave_item_ptr.data(), pl->wave_item_ptr.size() * 4, hipMemcpyHostToDevice, st);
            if (err != hipSuccess) return bail((int)err);
            const int serp = cfg->round_order == 2 ? 1 : 0;
            hipLaunchKernelGGL(k_dp_item_keys, dim3(ceil_div(n_pieces, kBlock)), dim3(kBlock), 0, st, ws.pieces, n_pieces, ws.sorted,
                               WPBR, NB, serp, ws.keys_a, ws.idx_a);
            size_t cb = ws.cub_bytes;
            int wave_bits = 1;
            while ((int64_t(1) << wave_bits) < n_waves) ++wave_bits;
            err = hipcub::DeviceRadixSort::SortPairs(ws.cub, cb, ws.keys_a, ws.keys_b, ws.idx_a, ws.idx_b, (int)ne, 0, 40 + wave_bits, st);
            if (err != hipSuccess) return bail((int)err);
            const uint32_t pad_x = 0x00FFFFFFu | ((uint32_t)CAP << 24);
            hipLaunchKernelGGL((k_dp_steps<false>), dim3(ceil_div(n_waves, kBlock / kWave)), dim3(kBlock), 0, st, ws.keys_b, ws.idx_b,
                               ws.sorted, ws.item_ptr, (int32_t)n_waves, GROUPS, cfg->lookahead, WPBR, NB, serp, pad_x, ws.steps, ws.pad,
                               (const int32_t *)nullptr, (uint32_t *)nullptr);
            err = hipMemcpyAsync(steps.data(), ws.steps, steps.size() * 4, hipMemcpyDeviceToHost, st);
            if (err == hipSuccess) err = hipMemcpyAsync(pads.data(), ws.pad, pads.size() * 4, hipMemcpyDeviceToHost, st);
            if (err == hipSuccess) err = hipStreamSynchronize(st);
            if (err != hipSuccess) return bail((int)err);
        }
        pl->wave_slab_ptr.assign((size_t)n_waves + 1, 0);
        int64_t n_slabs = 0, n_steps = 0, n_pad = 0;
        for (int64_t w = 0; w < n_waves; ++w) {
            pl->wave_slab_ptr[(size_t)w] = (int32_t)n_slabs;
            n_slabs += (steps[(size_t)w] + 31) / 32;
            n_steps += steps[(size_t)w];
            n_pad += pads[(size_t)w];
        }
        if (n_slabs >= INT32_MAX) return bail(LGC_E_RANGE);
        pl->wave_slab_ptr[(size_t)n_waves] = (int32_t)n_slabs;
        pl->keys = ws.keys_b;
        pl->idx = ws.idx_b;
        pl->sorted = ws.sorted;
        pl->d_item_ptr = ws.item_ptr;
        pl->d_slab_ptr = ws.slab_ptr;
        pl->dims.n_bands = NB;
        pl->dims.rounds = pp.rounds;
        pl->dims.row_cap = CAP;
        pl->dims.piece_cap = pp.PCAP;
        pl->dims.n_waves = n_waves;
        pl->dims.n_slabs = n_slabs;
        pl->dims.groups = GROUPS;
        pl->dims.n_slots = n_pieces;
        pl->dims.n_rows = (int32_t)n_rows;
        pl->dims.n_entries = ne;
        pl->dims.n_steps = n_steps;
        pl->dims.n_padding = n_pad;
        (void)SLAB;
    } catch (const std::bad_alloc &) {
        return bail((int)hipErrorOutOfMemory);
    }
    if (code) *code = 0;
    return pl;
}

int lgc_sweep_dplan_dims(const lgc_sweep_dplan *plan, lgc_sweep_dims *dims) {
    if (!plan || !dims) return LGC_E_INVAL;
    *dims = plan->dims;
    return 0;
}

int lgc_sweep_dplan_export_multi(const lgc_sweep_dplan *plan, lgc_multi_row *multi) {
    if (!plan || !multi) return LGC_E_INVAL;
    std::copy(plan->multi.begin(), plan->multi.end(), multi);
    return 0;
}

int lgc_sweep_dplan_fill(const lgc_sweep_dplan *plan, uint32_t *slabs, int32_t *wave_slab_ptr, int32_t *wave_npieces,
                         int32_t *piece_slot, void *stream_) {
    if (!plan || !slabs || !wave_slab_ptr || !wave_npieces || !piece_slot) return LGC_E_INVAL;
    hipStream_t st = as_stream(stream_);
    auto up = [&](void *dst, const void *src, size_t bytes) -> int {
        return bytes == 0 ? 0 : (int)hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st);
    };
    int rc = up(wave_slab_ptr, plan->wave_slab_ptr.data(), plan->wave_slab_ptr.size() * 4);
    if (rc == 0) rc = up(wave_npieces, plan->wave_npieces.data(), plan->wave_npieces.size() * 4);
    if (rc == 0) rc = up(piece_slot, plan->piece_slot.data(), plan->piece_slot.size() * 4);
    if (rc != 0) return rc;
    const lgc_sweep_cfg &cfg = plan->cfg;
    const int64_t n_waves = plan->dims.n_waves;
    if (n_waves > 0 && plan->dims.n_entries > 0) {
        const uint32_t pad_x = 0x00FFFFFFu | ((uint32_t)cfg.row_cap << 24);
        hipLaunchKernelGGL((k_dp_steps<true>), dim3(ceil_div(n_waves, kBlock / kWave)), dim3(kBlock), 0, st, plan->keys, plan->idx,
                           plan->sorted, plan->d_item_ptr, (int32_t)n_waves, plan->dims.groups, cfg.lookahead,
                           cfg.waves_per_band_round, cfg.n_bands, cfg.round_order == 2 ? 1 : 0, pad_x, (int32_t *)nullptr,
                           (int32_t *)nullptr, (const int32_t *)wave_slab_ptr, slabs);
    }
    rc = (int)hipGetLastError();
    if (rc != 0) return rc;
    return (int)hipStreamSynchronize(st);        // the workspace may be released as soon as this returns
}

void lgc_sweep_dplan_free(lgc_sweep_dplan *plan) { delete plan; }

// 61..64 columns: four table rows per gather instruction.  68..96: two (k_sweep_wide, 96-float accumulators, more rounds).
// 97..128: the four-row plan again, run twice -- columns [0, 64) and [64, dim) -- into one partial table; measured against
// the wide sweep this loses at 80 / 90 columns (916 vs 905, 970 vs 951 us per hop), wins at 96 (915 vs 937) and gives
// 128 columns a sweep at all (1130 vs 1326 us per hop with the chunked item step).
// A table the sweep kernels can address: 24-bit row ids, 32-bit byte offsets, and the padding id 0xFFFFFF out of range.
static bool sweep_table_ok(int32_t dim, int64_t table_rows, int64_t x_stride) {
    if (table_rows <= 0 || table_rows >= 0xFFFFFF || x_stride < dim) return false;
    const int64_t bytes = ((table_rows - 1) * x_stride + dim) * 4;
    const uint32_t pad = (uint32_t)(0xFFFFFFull * (uint64_t)(x_stride * 4));
    return x_stride * 4 < (1 << 24) && bytes < (int64_t(1) << 32) && (int64_t)pad >= bytes;
}

int lgc_sweep_ok(int32_t dim, int64_t table_rows, int64_t x_stride) {
    const int groups = (dim >= 61 && dim <= 64) ? 4 : (dim >= 68 && dim <= kWideRow) ? 2 : (dim > kWideRow && dim <= 128) ? 4 : 0;
    return groups != 0 && sweep_table_ok(dim, table_rows, x_stride) ? groups : 0;
}

int lgc_spmm_sweep(const uint32_t *slabs, const int32_t *wave_slab_ptr, const int32_t *wave_npieces, const int32_t *piece_slot,
                   int64_t n_waves, int32_t row_cap, int32_t groups, const lgc_multi_row *multi, int32_t n_rows,
                   const lgc_multi_row *multi_wide, int32_t n_wide, float *partials,
                   int64_t table_rows, const float *x, int64_t x_stride, float *y, int64_t y_stride, const float *r,
                   int64_t r_stride, float a, float b, int32_t dim, void *stream_) {
    if (!slabs || !wave_slab_ptr || !wave_npieces || !piece_slot || !partials || !x || !y || n_waves < 0 ||
        n_waves % 4 != 0 || n_rows < 0 || n_wide < 0 || (n_rows > 0 && !multi) || (n_wide > 0 && !multi_wide) || row_cap < 1 ||
        row_cap > 254 || x == y)
        return LGC_E_INVAL;
    if (groups == 0) groups = 4;
    if (lgc_sweep_ok(dim, table_rows, x_stride) != groups) return LGC_E_DIM;    // the plan's step width must fit the table
    if (y_stride < dim || (r && r_stride < dim)) return LGC_E_INVAL;
    if (!aligned_to(x, 4) || !aligned_to(y, 4) || (r && !aligned_to(r, 4)) || !aligned_to(slabs, 16) || !aligned_to(partials, 16))
        return LGC_E_ALIGN;
    hipStream_t stream = as_stream(stream_);
    const size_t lds = (size_t)(kBlock / kWave) * (size_t)(row_cap + 1) * (groups == 2 ? kWideRow : 64) * sizeof(float);
    if (lds > 160 * 1024) return LGC_E_INVAL;
    if (n_waves > 0) {
        SweepArgs p{reinterpret_cast<const u4 *>(slabs), wave_slab_ptr, wave_npieces, piece_slot, x, partials, x_stride,
                    (uint32_t)(((table_rows - 1) * x_stride + dim) * 4), dim, row_cap, (int32_t)n_waves, 0, nullptr, dim, 0};
        const bool two_pass = groups == 4 && dim > 64;   // columns [0, 64) and [64, dim) as two sweeps of the same plan
        if (two_pass) {
            p.dim = 64;
            p.x_bytes = (uint32_t)(((table_rows - 1) * x_stride + 64) * 4);
        }
#ifdef LGC_SWEEP_TRACE   /* debug builds only (tools/sweep_trace.py): the variable carries a device address */
        if (const char *tr = getenv("LGCN_SWEEP_TRACE")) p.trace = reinterpret_cast<unsigned long long *>(strtoull(tr, nullptr, 0));
#endif
        static unsigned long long lds_ok_sweep = 0, lds_ok_wide = 0;
        int rc_attr = allow_big_lds(reinterpret_cast<const void *>(k_sweep<kSweepDepth>), 160 * 1024, &lds_ok_sweep);
        if (rc_attr == 0) rc_attr = allow_big_lds(reinterpret_cast<const void *>(k_sweep_wide<8>), 160 * 1024, &lds_ok_wide);
        if (rc_attr != 0) return rc_attr;
        const int64_t req = knobs().sweep_launch_waves;
        const int64_t per_launch = req > 0 ? std::max<int64_t>(4, (req / 4) * 4) : n_waves;
        for (int64_t w0 = 0; w0 < n_waves; w0 += per_launch) {
            p.wave_begin = (int32_t)w0;
            const unsigned blocks = (unsigned)(std::min<int64_t>(per_launch, n_waves - w0) / 4);
            if (groups == 2) hipLaunchKernelGGL(k_sweep_wide<8>, dim3(blocks), dim3(kBlock), lds, stream, p);
            else hipLaunchKernelGGL(k_sweep<kSweepDepth>, dim3(blocks), dim3(kBlock), lds, stream, p);
            if (two_pass) {
                SweepArgs q = p;
                q.x = x + 64;
                q.dim = dim - 64;
                q.x_bytes = (uint32_t)(((table_rows - 1) * x_stride + (dim - 64)) * 4);
                q.pcol = 64;
                hipLaunchKernelGGL(k_sweep<kSweepDepth>, dim3(blocks), dim3(kBlock), lds, stream, q);
            }
        }
    }
    DimCfg cfg;
    dim_cfg(dim, &cfg);
    SpmmArgs sp{nullptr, nullptr, x, y, r, x_stride, y_stride, r_stride, a, b, dim, cfg.lpr, 0, 0, 0, 0};
    sp.wt_store = (table_rows * y_stride * 4 < (int64_t(1) << 32)) ? 1 : 0;
    if (n_wide + n_rows > 0) {
        const int rows_per_block = (kBlock / kWave) * (kWave / cfg.lpr);
        hipLaunchKernelGGL((k_sweep_combine<4>), dim3(n_wide + ceil_div(n_rows, rows_per_block)), dim3(kBlock), 0, stream, sp,
                           multi_wide, n_wide, multi, n_rows, partials);
    }
    return (int)hipGetLastError();
}

int lgc_apply(const lgc_operator *op, int64_t table_rows, const float *x, int64_t x_stride, float *y, int64_t y_stride,
              const float *r, int64_t r_stride, float a, float b, int32_t dim, void *stream) {
    if (!op || op->n_tile_classes < 0 || op->n_tile_classes > 3) return LGC_E_INVAL;
    if (op->sweep && lgc_sweep_ok(dim, table_rows, x_stride) == (op->sweep->groups ? op->sweep->groups : 4)) {
        const lgc_sweep_arrays *sw = op->sweep;
        return lgc_spmm_sweep(sw->slabs, sw->wave_slab_ptr, sw->wave_npieces, sw->piece_slot, sw->n_waves, sw->row_cap,
                              sw->groups, sw->multi, sw->n_rows, sw->multi_wide, sw->n_wide, sw->partials, table_rows, x, x_stride, y,
                              y_stride, r, r_stride, a, b, dim, stream);
    }
    const bool tiled = op->n_tile_classes > 0 && dim >= 4;
    if (tiled && !knobs().no_fused_apply) {
        // one launch: [chunk workgroups | tile classes], then the fixed-order combine of rows cut into several chunks
        DimCfg cfg;
        if (!dim_cfg(dim, &cfg)) return LGC_E_DIM;
        if (!x || !y || x == y || x_stride < dim || y_stride < dim || (r && r_stride < dim) || op->n_chunks < 0 || op->n_multi < 0 ||
            (op->n_chunks > 0 && (!op->chunks || !op->rowptr || !op->entries)) || (op->n_multi > 0 && (!op->multi || !op->partials)))
            return LGC_E_INVAL;
        FusedArgs f{};
        f.sp = SpmmArgs{op->rowptr, op->entries, x, y, r, x_stride, y_stride, r_stride, a, b, dim, cfg.lpr, op->row_begin,
                        op->row_begin, op->short_max, 0};
        f.sp.wt_store = (table_rows * y_stride * 4 < (int64_t(1) << 32)) ? 1 : 0;
        f.chunks = op->chunks;
        f.partials = op->partials;
        f.n_chunks = op->n_chunks;
        f.chunk_blocks = op->n_chunks > 0 ? ceil_div(op->n_chunks, kBlock / kWave) : 0;
        int64_t total = f.chunk_blocks;
        bool fast = true;
        for (int c = op->n_tile_classes - 1; c >= 0; --c) {   // widest class first: its wavefronts run longest (a rank of
            const lgc_tile_class &tc = op->tiles[c];          // an 8-way partition: 113 vs 116 us per hop)
            TilePrep tp;
            const int rc = prepare_tiles(tp, tc.order, tc.meta, tc.slab, tc.n_tiles, tc.width,
                                         op->tiles_per_wave > 0 ? op->tiles_per_wave : 1, table_rows, x, x_stride, y, y_stride,
                                         r, r_stride, a, b, dim);
            if (rc != 0) return rc;
            if (tc.n_tiles == 0) continue;
            fast = fast && tp.fast;
            f.t[f.n_classes] = tp.p;
            f.width[f.n_classes] = tc.width;
            f.blocks[f.n_classes] = (int32_t)tp.blocks;
            total += tp.blocks;
            ++f.n_classes;
        }
        hipStream_t st = as_stream(stream);
        if (total > 0) {
            if (fast && dim > 64) hipLaunchKernelGGL((k_apply_fused<2>), dim3((unsigned)total), dim3(kBlock), 0, st, f);
            else if (fast) hipLaunchKernelGGL((k_apply_fused<1>), dim3((unsigned)total), dim3(kBlock), 0, st, f);
            else hipLaunchKernelGGL((k_apply_fused<0>), dim3((unsigned)total), dim3(kBlock), 0, st, f);
        }
        if (op->n_multi > 0)
            hipLaunchKernelGGL((k_spmm_combine<4>), dim3(ceil_div(op->n_multi, kBlock / kWave)), dim3(kBlock), 0, st, f.sp, op->multi,
                               op->n_multi, op->partials);
        return (int)hipGetLastError();
    }
    if (!tiled || op->n_chunks > 0) {   // long rows first (they run longest); with tiles the row part gets an empty range
        const int rc = lgc_spmm(op->rowptr, op->entries, op->row_begin, tiled ? op->row_begin : op->row_end, op->short_max,
                                op->chunks, op->n_chunks, op->multi, op->n_multi, op->partials, table_rows, x, x_stride, y,
                                y_stride, r, r_stride, a, b, dim, stream);
        if (rc != 0) return rc;
    }
    if (tiled) {
        for (int c = 0; c < op->n_tile_classes; ++c) {
            const lgc_tile_class &tc = op->tiles[c];
            const int rc = lgc_spmm_tiles(tc.order, tc.meta, tc.slab, tc.n_tiles, tc.width,
                                          op->tiles_per_wave > 0 ? op->tiles_per_wave : 1, table_rows, x, x_stride, y,
                                          y_stride, r, r_stride, a, b, dim, stream);
            if (rc != 0) return rc;
        }
    }
    return 0;
}

int lgc_hop_exchange(const lgc_operator *item_op, const lgc_operator *user_op, int64_t table_rows, const float *x,
                     int64_t x_stride, float *y, int64_t y_stride, const float *r, int64_t r_stride, float a, float b,
                     int32_t dim, int32_t exchange_row_begin, int32_t exchange_rows, int32_t item_epilogue,
                     lgc_exchange_fn exchange, void *user, void *stream) {
    if (!item_op || !user_op || !exchange || exchange_row_begin < 0 || exchange_rows < 0 ||
        (int64_t)exchange_row_begin + exchange_rows > table_rows)
        return LGC_E_INVAL;
    // the exchanged block is SUMMED over ranks: the b * r term of the item rows may enter that sum once only
    int rc = lgc_apply(item_op, table_rows, x, x_stride, y, y_stride, item_epilogue ? r : nullptr, item_epilogue ? r_stride : 0, a,
                       item_epilogue ? b : 0.0f, dim, stream);
    if (rc != 0) return rc;
    rc = exchange(y + (int64_t)exchange_row_begin * y_stride, exchange_rows, y_stride, dim, stream, user);
    if (rc != 0) return rc;
    return lgc_apply(user_op, table_rows, x, x_stride, y, y_stride, r, r_stride, a, b, dim, stream);
}

int lgc_segment_sum(const int64_t *key_sorted, const int64_t *dest, const float *vals, const int32_t *vals_index, int64_t n,
                    float scale, float *y, int64_t y_stride, int64_t y_rows, int32_t dim, int32_t accumulate, void *stream_) {
    if (!y || n < 0 || y_rows < 0 || dim < 1 || dim > 256 || y_stride < dim) return LGC_E_INVAL;
    if (n == 0) return 0;
    if (!key_sorted || !dest || !vals) return LGC_E_INVAL;
    const int groups = kWave / ((dim + 3) / 4);
    hipLaunchKernelGGL(k_segment_sum, dim3(ceil_div(n, (int64_t)(kBlock / kWave) * groups)), dim3(kBlock), 0, as_stream(stream_),
                       key_sorted, dest, vals, vals_index, n, scale, y, y_stride, y_rows, dim, accumulate);
    return (int)hipGetLastError();
}

int lgc_seed_prepare(const int64_t *rows, int64_t m, int64_t split, int64_t n_nodes, int64_t *rows_sorted, int32_t *perm,
                     int64_t *dest_item, int64_t *dest_slot, int64_t *dest_user, uint8_t *col_flag, int32_t *col_slot,
                     uint64_t *scratch, void *stream_) {
    if (m < 0 || m > LGC_SEED_MAX) return LGC_E_RANGE;
    if (split < 0 || n_nodes < split || (col_flag != nullptr) != (col_slot != nullptr)) return LGC_E_INVAL;
    if (m == 0) return 0;
    if (!rows || !rows_sorted || !perm || !dest_item || !dest_slot || !dest_user || !scratch) return LGC_E_INVAL;
    hipStream_t st = as_stream(stream_);
    hipLaunchKernelGGL(k_seed_rank, dim3(ceil_div(m, kRankKeys)), dim3(kBlock), 0, st, rows, (int32_t)m, n_nodes,
                       reinterpret_cast<unsigned long long *>(scratch));
    hipLaunchKernelGGL(k_seed_finish, dim3(ceil_div(m, kBlock)), dim3(kBlock), 0, st,
                       reinterpret_cast<const unsigned long long *>(scratch), (int32_t)m, split, rows_sorted, perm, dest_item,
                       dest_slot, dest_user, col_flag, col_slot);
    return (int)hipGetLastError();
}

int lgc_seed_flags(const int64_t *rows_sorted, int64_t m, int64_t split, uint8_t *col_flag, int32_t value, void *stream_) {
    if (m < 0 || split < 0 || value < 0 || value > 255) return LGC_E_INVAL;
    if (m == 0) return 0;
    if (!rows_sorted || !col_flag) return LGC_E_INVAL;
    hipLaunchKernelGGL(k_seed_flags, dim3(ceil_div(m, kBlock)), dim3(kBlock), 0, as_stream(stream_), rows_sorted, m, split,
                       col_flag, (uint8_t)value);
    return (int)hipGetLastError();
}

int lgc_pair_dot_rows(const float *emb, int64_t stride, int32_t dim, int64_t n_nodes, const int64_t *idx0, const int64_t *idx1,
                      int64_t n_pairs, float *scores, float *rows0, float *rows1, uint8_t *ok, int32_t *status, void *stream_) {
    if (!emb || !status || dim < 1 || stride < dim || n_nodes < 0 || n_pairs < 0) return LGC_E_INVAL;
    if (n_pairs == 0) return 0;
    if (!idx0 || !idx1 || !scores) return LGC_E_INVAL;
    hipLaunchKernelGGL(k_pair_dot_rows, dim3(ceil_div(n_pairs, kBlock / kWave)), dim3(kBlock), 0, as_stream(stream_), emb, stride,
                       dim, n_nodes, idx0, idx1, n_pairs, scores, rows0, rows1, ok, status);
    return (int)hipGetLastError();
}

int lgc_pair_seed_vals(const float *grad_scores, const uint8_t *mask, const float *grad_scale, const float *rows0,
                       const float *rows1, int64_t n_pairs, int32_t dim, float *vals, void *stream_) {
    if (n_pairs < 0 || dim < 1) return LGC_E_INVAL;
    if (n_pairs == 0) return 0;
    if (!grad_scores || !rows0 || !rows1 || !vals) return LGC_E_INVAL;
    hipLaunchKernelGGL(k_pair_seed_vals, dim3(ceil_div(n_pairs * dim, kBlock)), dim3(kBlock), 0, as_stream(stream_), grad_scores,
                       mask, grad_scale, rows0, rows1, n_pairs, dim, vals);
    return (int)hipGetLastError();
}

int lgc_bpr_loss(const float *scores, const uint8_t *mask, int64_t n_triples, int64_t size, float *loss, float *grad,
                 void *stream_) {
    if (!loss || n_triples < 0 || size <= 0) return LGC_E_INVAL;
    if (n_triples > 0 && (!scores || !grad)) return LGC_E_INVAL;
    hipLaunchKernelGGL(k_bpr_loss, dim3(1), dim3(kSeedBlock), 0, as_stream(stream_), scores, mask, n_triples,
                       1.0f / (float)size, loss, grad);
    return (int)hipGetLastError();
}

int lgc_reg_rows(const float *w, int64_t stride, int32_t dim, int64_t n_rows, const int64_t *ids0, int64_t m0, const int64_t *ids1,
                 int64_t m1, const int64_t *ids2, int64_t m2, float scale, float *value, int64_t *rows_out, int32_t *status,
                 void *stream_) {
    if (!w || !value || !status || dim < 1 || stride < dim || n_rows < 0 || m0 < 0 || m1 < 0 || m2 < 0) return LGC_E_INVAL;
    if ((m0 > 0 && !ids0) || (m1 > 0 && !ids1) || (m2 > 0 && !ids2)) return LGC_E_INVAL;
    hipLaunchKernelGGL(k_reg_rows, dim3(1), dim3(kSeedBlock), 0, as_stream(stream_), w, stride, dim, n_rows, ids0, m0, ids1, m1, ids2,
                       m2, scale, value, rows_out, status);
    return (int)hipGetLastError();
}

int lgc_seed_mark(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end, const int64_t *seed_rows,
                  int64_t n_seed, uint8_t *mark, int64_t mark_len, int32_t value, void *stream_) {
    if (n_seed < 0 || row_begin < 0 || row_end < row_begin || mark_len < 0 || value < 0 || value > 255) return LGC_E_INVAL;
    if (n_seed == 0) return 0;
    if (!rowptr || !entries || !seed_rows || !mark) return LGC_E_INVAL;
    hipLaunchKernelGGL(k_seed_mark, dim3(ceil_div(n_seed, (int64_t)(kBlock / kWave))), dim3(kBlock), 0, as_stream(stream_), rowptr, entries,
                       row_begin, row_end, seed_rows, n_seed, mark, mark_len, (uint8_t)value);
    return (int)hipGetLastError();
}

int lgc_seed_pull(const int32_t *rowptr, const lgc_entry *entries, int32_t row_begin, int32_t row_end, int32_t short_max,
                  const lgc_chunk *chunks, int32_t n_chunks, const lgc_multi_row *multi, int32_t n_multi, float *partials,
                  const uint8_t *col_flag, const int32_t *col_slot, const uint8_t *row_mark, const float *seed_vals,
                  int64_t seed_stride, int64_t table_rows, float *y, int64_t y_stride, int32_t dim, void *stream_) {
    DimCfg cfg;
    if (!dim_cfg(dim, &cfg)) return LGC_E_DIM;
    if (!rowptr || !col_flag || !col_slot || !seed_vals || !y || row_begin < 0 || row_end < row_begin || n_chunks < 0 ||
        n_multi < 0 || short_max < 0 || table_rows < row_end || seed_stride < dim || y_stride < dim)
        return LGC_E_INVAL;
    if ((n_chunks > 0 && !chunks) || (n_multi > 0 && (!multi || !partials))) return LGC_E_INVAL;
    if (!aligned_to(seed_vals, 4) || !aligned_to(y, 4)) return LGC_E_ALIGN;
    hipStream_t stream = as_stream(stream_);
    SpmmArgs p{rowptr, entries, seed_vals, y, nullptr, seed_stride, y_stride, 0, 1.0f, 0.0f, dim, cfg.lpr, row_begin, row_end,
               short_max, 0, col_flag, col_slot, row_mark};
    const int waves_per_block = kBlock / kWave;
    const int rows_per_wave = kWave / cfg.lpr;
    return dispatch_dim(cfg, [&](auto vec) -> int {
        constexpr int V = decltype(vec)::value;
        const int64_t n_rows = (int64_t)row_end - row_begin;
        const int row_blocks = n_rows > 0 ? ceil_div(ceil_div(n_rows, rows_per_wave), waves_per_block) : 0;
        const int chunk_blocks = n_chunks > 0 ? ceil_div(n_chunks, waves_per_block) : 0;
        if (row_mark != nullptr && n_chunks > 0) {
            if (row_blocks > 0)
                hipLaunchKernelGGL((k_spmm_hop<V, true>), dim3(row_blocks), dim3(kBlock), 0, stream, p, chunks, 0, partials, 0);
            hipLaunchKernelGGL((k_seed_pull_chunks<V>), dim3(ceil_div(ceil_div(n_chunks, kScreen), waves_per_block)), dim3(kBlock), 0,
                               stream, p, chunks, n_chunks, partials);
        } else if (row_blocks + chunk_blocks > 0) {
            hipLaunchKernelGGL((k_spmm_hop<V, true>), dim3(row_blocks + chunk_blocks), dim3(kBlock), 0, stream, p, chunks, n_chunks,
                               partials, chunk_blocks);
        }
        if (n_multi > 0)
            hipLaunchKernelGGL((k_spmm_combine<V>), dim3(ceil_div(n_multi, waves_per_block)), dim3(kBlock), 0, stream, p, multi,
                               n_multi, partials);
        return (int)hipGetLastError();
    });
}

int lgc_lincomb(float *y, int64_t y_stride, const float *const *src, const int64_t *src_stride, const float *coef,
                int32_t n_terms, int64_t n_rows, int32_t dim, void *stream_) {
    if (!y || !src || !src_stride || !coef || n_terms < 1 || n_terms > LGC_MAX_TERMS || n_rows < 0 || dim < 1 ||
        y_stride < dim)
        return LGC_E_INVAL;
    LincombArgs a{};
    a.n_terms = n_terms;
    for (int t = 0; t < n_terms; ++t) {
        if (!src[t] || src_stride[t] < dim) return LGC_E_INVAL;
        a.src[t] = src[t];
        a.stride[t] = src_stride[t];
        a.coef[t] = coef[t];
    }
    if (n_rows == 0) return 0;
    int blocks = (int)std::min<int64_t>(ceil_div(n_rows * dim, kBlock), 256 * 8);
    hipLaunchKernelGGL(k_lincomb, dim3(blocks), dim3(kBlock), 0, as_stream(stream_), y, y_stride, a, n_rows, dim);
    return (int)hipGetLastError();
}

static int adam_launch(float *w, const float *g, float *m, float *v, int64_t n, float one_minus_beta1, float beta2,
                       float one_minus_beta2, float eps, float step_size, float bias_correction2_sqrt, const float *hyper,
                       void *stream_);

int lgc_adam_step(float *w, const float *g, float *m, float *v, int64_t n, float one_minus_beta1, float beta2,
                  float one_minus_beta2, float eps, float step_size, float bias_correction2_sqrt, void *stream_) {
    if (!(bias_correction2_sqrt > 0.0f)) return LGC_E_INVAL;
    return adam_launch(w, g, m, v, n, one_minus_beta1, beta2, one_minus_beta2, eps, step_size, bias_correction2_sqrt, nullptr,
                       stream_);
}

int lgc_adam_step_hp(float *w, const float *g, float *m, float *v, int64_t n, const float *hyper, void *stream_) {
    if (!hyper) return LGC_E_INVAL;
    return adam_launch(w, g, m, v, n, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 1.0f, hyper, stream_);
}

static int adam_launch(float *w, const float *g, float *m, float *v, int64_t n, float one_minus_beta1, float beta2,
                       float one_minus_beta2, float eps, float step_size, float bias_correction2_sqrt, const float *hyper,
                       void *stream_) {
    if (!w || !g || !m || !v || n < 0) return LGC_E_INVAL;
    // dword-aligned, and all four at the same offset inside a 16-byte line (a row range of same-shaped tables whose rows
    // are not whole float4s, e.g. rows [lo, hi) of a [N, 90] table): the first elements up to the line are done one by one
    const uintptr_t mis = reinterpret_cast<uintptr_t>(w) & 15;
    if ((mis & 3) != 0 || (reinterpret_cast<uintptr_t>(g) & 15) != mis || (reinterpret_cast<uintptr_t>(m) & 15) != mis ||
        (reinterpret_cast<uintptr_t>(v) & 15) != mis)
        return LGC_E_ALIGN;
    if (n == 0) return 0;
    hipStream_t stream = as_stream(stream_);
    const int64_t head = std::min<int64_t>(n, (int64_t)((16 - mis) & 15) / 4);
    if (head > 0)
        hipLaunchKernelGGL(k_adam, dim3(1), dim3(kBlock), 0, stream, w, g, m, v, (int64_t)0, head, beta2, one_minus_beta1,
                           one_minus_beta2, eps, step_size, bias_correction2_sqrt, hyper);
    w += head; g += head; m += head; v += head; n -= head;
    if (n == 0) return (int)hipGetLastError();
    const int64_t n4 = n / 4;
    const int64_t blocks = std::max<int64_t>(ceil_div(n4, (int64_t)kBlock * kAdamU), 1);
    if (blocks >= INT32_MAX) return LGC_E_RANGE;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(kBlock), 0, stream, w, g, m, v, n4, n, beta2,
                       one_minus_beta1, one_minus_beta2, eps, step_size, bias_correction2_sqrt, hyper);
    return (int)hipGetLastError();
}

int lgc_pair_dot(const float *emb, int64_t stride, int32_t dim, int64_t n_nodes, const int64_t *idx0,
                 const int64_t *idx1, int64_t n_pairs, float *scores, int32_t *status, void *stream_) {
    if (!emb || !status || dim < 1 || stride < dim || n_nodes < 0 || n_pairs < 0) return LGC_E_INVAL;
    if (n_pairs == 0) return 0;
    if (!idx0 || !idx1 || !scores) return LGC_E_INVAL;
    hipLaunchKernelGGL(k_pair_dot, dim3(ceil_div(n_pairs, kBlock / kWave)), dim3(kBlock), 0, as_stream(stream_), emb,
                       stride, dim, n_nodes, idx0, idx1, n_pairs, scores, status);
    return (int)hipGetLastError();
}

int lgc_mask_topk(const float *scores, int64_t score_stride, const float *seen, int64_t seen_stride, const int64_t *list_ptr,
                  const int64_t *list_items, const int64_t *list_rows, int64_t n_rows, int32_t n_cols, int32_t k,
                  int64_t *out_index, float *out_value, void *stream_) {
    if (!scores || !out_index || n_rows < 0 || n_cols < 1 || k < 1 || k > n_cols || score_stride < n_cols ||
        (seen && seen_stride < n_cols) || n_rows >= INT32_MAX || (seen && list_ptr) || (list_ptr && !list_items))
        return LGC_E_INVAL;
    if (k > kTopkMax) return LGC_E_RANGE;
    const size_t lds = list_ptr ? (size_t)((n_cols + 31) / 32) * 4 : 0;
    if (lds > 120 * 1024) return LGC_E_RANGE;             // the bitmask form holds up to 983,040 columns
    if (n_rows == 0) return 0;
    const bool regs = n_cols <= kTopkRegs * kTopkBlock;
    const int mode = list_ptr ? 2 : seen ? 1 : 0;
    using topk_fn = void (*)(const float *, int64_t, const float *, int64_t, const int64_t *, const int64_t *,
                             const int64_t *, int32_t, int32_t, int64_t *, float *);
    static const topk_fn kerns[2][3] = {{k_mask_topk<false, 0>, k_mask_topk<false, 1>, k_mask_topk<false, 2>},
                                        {k_mask_topk<true, 0>, k_mask_topk<true, 1>, k_mask_topk<true, 2>}};
    const topk_fn kern = kerns[regs][mode];
    if (lds > 16 * 1024) {    // static LDS (histogram copies, candidates) + the bitmask can pass the 64 KiB default
        static unsigned long long lds_ok_topk[2][3] = {};
        const int rc_attr = allow_big_lds(reinterpret_cast<const void *>(kern), 120 * 1024, &lds_ok_topk[regs][mode]);
        if (rc_attr != 0) return rc_attr;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)n_rows), dim3(kTopkBlock), lds, as_stream(stream_), scores, score_stride, seen,
                       seen_stride, list_ptr, list_items, list_rows, n_cols, k, out_index, out_value);
    return (int)hipGetLastError();
}

int lgc_sample_triples(const int64_t *users, int64_t n, const int32_t *pos_ptr, const int64_t *pos_items,
                       const int32_t *ign_ptr, const int64_t *ign_items, int64_t n_users, int64_t n_items, uint64_t seed,
                       uint64_t step, int64_t *pos_out, int64_t *neg_out, int32_t *status, void *stream_) {
    if (n < 0 || n_users < 0 || n_items < 1 || !status) return LGC_E_INVAL;
    if (n == 0) return 0;
    if (!users || !pos_ptr || !pos_items || !ign_ptr || !pos_out || !neg_out) return LGC_E_INVAL;
    hipLaunchKernelGGL(k_sample_triples, dim3(ceil_div(n, kBlock)), dim3(kBlock), 0, as_stream(stream_), users, n,
                       pos_ptr, pos_items, ign_ptr, ign_items, n_users, n_items, seed, step, pos_out, neg_out, status);
    return (int)hipGetLastError();
}

#ifdef LGC_TOPK_TRACE
int lgc_debug_topk_trace(unsigned long long *out16) {
    return (int)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_topk_trace), sizeof(unsigned long long) * 16);
}
#endif

}  // extern "C"
