// cgic_common.h -- shared host/device helpers of libcgic_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <functional>
#include <vector>

#include "../../include/cgic_hip.h"

namespace cgic {

constexpr int kWave = 64;  // gfx950 wavefront width

// ---- host-side error plumbing ------------------------------------------------
void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define CGIC_HIP_TRY(expr)                                                      \
    do {                                                                        \
        hipError_t _e = (expr);                                                 \
        if (_e != hipSuccess) return ::cgic::hip_fail(_e, #expr, __FILE__, __LINE__); \
    } while (0)

#define CGIC_REQUIRE(cond, code, ...)                                           \
    do {                                                                        \
        if (!(cond)) {                                                          \
            ::cgic::set_error(__VA_ARGS__);                                     \
            return (code);                                                      \
        }                                                                       \
    } while (0)

inline int launch_check(const char *kernel)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("launch of %s failed: %s", kernel, hipGetErrorString(e));
        return CGIC_ERR_HIP;
    }
    return CGIC_OK;
}

constexpr int kLutBitsMax = 13;   // decode LUT width (entries of 4 bytes: 32 KB of LDS in the decode kernels)

// ---- device-side table view ----------------------------------------------------
struct TableDev {
    const int32_t *len;    // [n] code length in bits
    const uint32_t *code;  // [n * words] MSB-first code words
    const uint32_t *lut;   // [1 << lut_bits] decode LUT (see cgic_table.hip)
    const int32_t *child;  // [2 * nodes] decode trie: >=0 node id, <0 = ~symbol
    int n;
    int words;
    int max_len;
    int lut_bits;
    int n_nodes;      // decode trie nodes (child has 2 * n_nodes entries)
    int dbl_rounds;   // pointer-doubling rounds that cover a 64-bit chunk: ceil(log2(ceil(64 / min_len)))
    int len_gcd;      // gcd of all code lengths: every codeword boundary of a stream is a multiple of it (a table of equal
                      // lengths never re-synchronises from a wrong offset: the self-synchronising decoder guesses inside the
                      // right residue class)
};

// `n` consecutive self-resetting ticket words (zero on entry; the kernel that uses one must leave it zero),
// 64 bytes apart: ptr[0], ptr[16], ptr[32] ... (kTicketStride words).  See cgic_table.hip.
constexpr int kTicketStride = 16;
// kind 1: slots of a second pool with the router's refinement-queue contract instead (cgic_router_dev.h: zero when first handed out,
// afterwards whatever a finished launch left -- a queue word that says "nothing to claim", counters that only ever grow or return
// to zero): never mixed with the zero-on-entry slots of kind 0.
int acquire_tickets(hipStream_t stream, int n, unsigned int **ptr, int kind = 0);

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (function, device) and size -- not on every launch
int ensure_dynamic_lds(const void *fn, size_t bytes);

// ---- launch groups (cgic_group_begin / _select / _launch, cgic_launch.hip) ------------------------------------------
// Independent sub-batches of DIFFERENT shapes (the shape groups of one tiled image: inference_high_resolution.py:112-125 cuts
// a 2040x1356 image into six tiles of four shapes, :246 runs them one by one) go through ONE launch per kernel: between
// cgic_group_begin and cgic_group_launch the entry points of this thread record what they would have launched -- the host-side
// shape logic of every call runs unchanged -- and cgic_group_launch issues, position by position, one launch whose grid is the
// concatenation of the recorded grids.  A workgroup finds its group from blockIdx.x (<= kMaxGroups scalar compares), takes
// that group's argument block from the kernarg segment (scalar loads with a uniform offset) and runs the unchanged kernel
// body on its RELATIVE block index.  Kernels without a grouped form, or positions whose groups recorded different kernels,
// are launched one by one: same results.
constexpr int kMaxGroups = 4;           // shape groups of a tiled image: (full | ragged) x (full | ragged)
struct Blk { unsigned int x, y, z, nx, ny; };      // block index inside the group's own grid (+ that grid's x / y extents)
template <class A>
struct Grouped {
    unsigned int n;
    unsigned int start[kMaxGroups + 1];           // first flat block of each group; start[n] = total
    unsigned int gx[kMaxGroups], gy[kMaxGroups];  // the group's grid extents (x fastest, like the hardware's dispatch order)
    A a[kMaxGroups];
};
enum KernelId { KID_NONE = 0, KID_ENTROPY_F32, KID_ENTROPY_U8, KID_ENTROPY_WIN_F32, KID_ENTROPY_WIN_U8, KID_VQF_ROUTER_AL, KID_VQF_ROUTER_UN, KID_COMPRESS, KID_DECODE_SPLIT,
                KID_DECODE_IMAGE, KID_MERGE, KID_DECODE_MERGE, KID_COUNT };

// host side of a launch group
struct GroupRec {                       // one recorded launch
    int kid;
    hipStream_t stream;                 // the stream the entry point was called with (cgic_group_launch's must be the same one)
    dim3 grid, block;
    size_t lds;
    std::vector<unsigned char> args;    // the kernel's argument block (the A of Grouped<A>)
    std::function<int()> launch;        // the launch as it would have been made (fallback)
};
bool group_recording();                 // this thread is between cgic_group_begin and cgic_group_launch
double group_cu_share();                // the current group's share of the chip (1.0 outside a group): persistent-workgroup kernels size their grid by it
int group_record(int kid, dim3 grid, dim3 block, size_t lds, const void *args, size_t bytes, hipStream_t stream, std::function<int()> launch);
typedef int (*GroupedLauncher)(const GroupRec *const *recs, int n, hipStream_t s);
struct GroupedRegistrar { GroupedRegistrar(int kid, GroupedLauncher fn); };

// entry points without a recorded form refuse to run inside a group (their launch would overtake the recorded ones)
#define CGIC_NOT_IN_GROUP(name) CGIC_REQUIRE(!::cgic::group_recording(), CGIC_ERR_INVALID, name ": not available between cgic_group_begin and cgic_group_launch (it has no recorded form: call it before or after the group)")

// launch now, or record for cgic_group_launch
template <class A, class F>
inline int launch_or_record(int kid, dim3 grid, dim3 block, size_t lds, const A &a, hipStream_t stream, F direct)
{
    if (group_recording()) return group_record(kid, grid, block, lds, &a, sizeof(A), stream, std::function<int()>(direct));
    return direct();
}

// the argument block of a grouped launch from the records of one position (same kernel id, same block size: checked by the caller)
template <class A>
inline int fill_grouped(const GroupRec *const *recs, int n, Grouped<A> *g, size_t *lds_max)
{
    static_assert(sizeof(Grouped<A>) <= 4096, "the grouped argument block must fit the kernarg segment");
    CGIC_REQUIRE(n >= 1 && n <= kMaxGroups, CGIC_ERR_INVALID, "group launch: %d groups", n);
    g->n = (unsigned int)n;
    unsigned int at = 0;
    *lds_max = 0;
    for (int i = 0; i < kMaxGroups; ++i) {
        const GroupRec *r = recs[i < n ? i : n - 1];          // unused entries repeat the last group (never selected)
        CGIC_REQUIRE(r->args.size() == sizeof(A), CGIC_ERR_INVALID, "group launch: argument block of %zu bytes, expected %zu", r->args.size(), sizeof(A));
        memcpy((void *)&g->a[i], r->args.data(), sizeof(A));
        g->gx[i] = r->grid.x; g->gy[i] = r->grid.y;
        g->start[i] = at;
        if (i < n) {
            at += r->grid.x * r->grid.y * r->grid.z;
            if (r->lds > *lds_max) *lds_max = r->lds;
        }
    }
    for (int i = n; i <= kMaxGroups; ++i) g->start[i] = at;
    return CGIC_OK;
}

struct Table;  // host object behind cgic_table
int table_device_view(const cgic_table *t, TableDev *out);  // uploads lazily

#if defined(__HIPCC__)
// Debug-only phase stamps (make dbg -> libcgic_hip_dbg.so, -DCGIC_PHASE_CLOCKS): workgroup 0 /
// thread 0 records the shader clock at phase boundaries.  Compiled out of the product library.
#ifdef CGIC_PHASE_CLOCKS
extern __device__ long long g_phase_clk[32];
extern __device__ long long g_blk_t[2 * 4096];   // per-workgroup (start, end) on the 100 MHz clock
#define CGIC_BLK_BEGIN() do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_blk_t[2 * blockIdx.x] = wall_clock64(); } while (0)
#define CGIC_BLK_END() do { if (threadIdx.x == 0 && blockIdx.x < 4096) g_blk_t[2 * blockIdx.x + 1] = wall_clock64(); } while (0)
// dbg counters: slot 2 * (2048 + workgroup) (+1) of g_blk_t counts events of that workgroup (launches of <= 2048 workgroups)
#define CGIC_DBG_COUNT(which, n) do { if (lane_id() == 0 && blockIdx.x < 2048) atomicAdd((unsigned long long *)&g_blk_t[2 * (2048 + blockIdx.x) + (which)], (unsigned long long)(n)); } while (0)
#define CGIC_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) g_phase_clk[i] = clock64(); \
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 192 && (i) >= 2 && (i) <= 7) g_phase_clk[8 + (i)] = clock64(); } while (0)
// per-phase accumulation over the groups of a wave (workgroup 0: wave 0 -> g_phase_clk[16 + k], wave 4 -> [22 + k]; k < 6)
#define CGIC_PHASE_T0() long long _ph_t = clock64()
#define CGIC_PHASE_ACC(k) do { if (blockIdx.x == 0 && (threadIdx.x == 0 || threadIdx.x == 256)) { const long long _n = clock64(); g_phase_clk[(threadIdx.x ? 22 : 16) + (k)] += _n - _ph_t; _ph_t = _n; } } while (0)
// span of a whole launch over ALL workgroups (constant 100 MHz clock): [28] = earliest start, [29] = latest end,
// [30]/[31] = start/end of the workgroup that ended last (as block id pair packed)
#define CGIC_SPAN_BEGIN() long long _span_t0 = 0; do { if (threadIdx.x == 0) { _span_t0 = wall_clock64(); atomicMin((unsigned long long *)&g_phase_clk[28], (unsigned long long)_span_t0); } } while (0)
#define CGIC_SPAN_END() do { if (threadIdx.x == 0) { long long _t1 = wall_clock64(); unsigned long long _old = atomicMax((unsigned long long *)&g_phase_clk[29], (unsigned long long)_t1); if ((unsigned long long)_t1 > _old) { g_phase_clk[30] = _t1 - _span_t0; g_phase_clk[31] = blockIdx.x + 1000 * blockIdx.y; } } } while (0)
// STAMP3: the medium-stream workgroup of image 0 in the (B, 3) decode grid (medium = row 0), wave 0
#define CGIC_STAMP3(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_phase_clk[i] = clock64(); } while (0)
// STAMP2: the fine-stream workgroup of image 0 in the (B, jobs) compress grid (fine = row 0)
#define CGIC_STAMP2(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) g_phase_clk[i] = clock64(); } while (0)
#else
#define CGIC_STAMP(i) do {} while (0)
#define CGIC_PHASE_T0() do {} while (0)
#define CGIC_PHASE_ACC(k) do {} while (0)
#define CGIC_DBG_COUNT(which, n) do {} while (0)
#define CGIC_STAMP2(i) do {} while (0)
#define CGIC_STAMP3(i) do {} while (0)
#define CGIC_BLK_BEGIN() do {} while (0)
#define CGIC_BLK_END() do {} while (0)
#define CGIC_SPAN_BEGIN() do {} while (0)
#define CGIC_SPAN_END() do {} while (0)
#endif

// ---- wave / block primitives ---------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & (kWave - 1)); }

// which group this workgroup belongs to and its block index inside that group's grid (all scalar: blockIdx + kernarg loads)
template <class A>
__device__ __forceinline__ int group_locate(const Grouped<A> &g, Blk *blk)
{
    const unsigned int b = blockIdx.x;
    int i = 0;
#pragma unroll
    for (int k = 1; k < kMaxGroups; ++k)
        if (b >= g.start[k]) i = k;            // (start[k] = total for k >= n: never reached)
    const unsigned int rel = b - g.start[i], nx = g.gx[i], ny = g.gy[i];
    const unsigned int t = rel / nx;
    blk->x = rel - t * nx; blk->z = t / ny; blk->y = t - blk->z * ny; blk->nx = nx; blk->ny = ny;
    return i;
}
__device__ __forceinline__ Blk own_blk() { return Blk{blockIdx.x, blockIdx.y, blockIdx.z, gridDim.x, gridDim.y}; }

// gfx950 v_permlane16_swap / v_permlane32_swap: exchange between the 16-lane rows / 32-lane halves of a wave on
// the VALU, no LDS round trip.  Called with a == b == x they return
//   swap16: a = x of rows (0,0,2,2), b = x of rows (1,1,3,3)      swap32: a = (lower half, lower half), b = (upper, upper)
// Inline asm on purpose: through __builtin_amdgcn_permlane16_swap hipcc (ROCm 7.2) was seen to fold r[0] + r[1]
// of one swap into 2 * r[0] (tools/probes/probe_dpp.hip).  s_nop 1 = the wait states after the VALU write of an operand.
__device__ __forceinline__ void swap16(unsigned int &a, unsigned int &b)
{
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap32(unsigned int &a, unsigned int &b)
{
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

// inclusive scan across the 64 lanes of a wave (Hillis-Steele over DPP-able shuffles)
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < kWave; d <<= 1) {
        T o = __shfl_up(v, d, kWave);
        if (lane >= d) v += o;
    }
    return v;
}

// The same for 32-bit integers on DPP operands: four row_shr steps inside the rows of 16 lanes, then row_bcast:15 / row_bcast:31
// carry the row totals over -- six dependent VALU instructions instead of six LDS-crossbar round trips (ds_bpermute) with a
// compare and a select each.  Matters where ONE wave's dependent chain is the critical path (the router's histogram scans in the
// fused VQ + router launch, whose waves only get the issue slots the VQ workgroup leaves over).
__device__ __forceinline__ unsigned int wave_inclusive_scan_u32(unsigned int v)
{
#define CGIC_DPP_ADD(ctrl, rmask) v += (unsigned int)__builtin_amdgcn_update_dpp(0, (int)v, ctrl, rmask, 0xF, false)
    CGIC_DPP_ADD(0x111, 0xF);        // row_shr:1
    CGIC_DPP_ADD(0x112, 0xF);        // row_shr:2
    CGIC_DPP_ADD(0x114, 0xF);        // row_shr:4
    CGIC_DPP_ADD(0x118, 0xF);        // row_shr:8
    CGIC_DPP_ADD(0x142, 0xA);        // row_bcast:15 into rows 1 and 3
    CGIC_DPP_ADD(0x143, 0xC);        // row_bcast:31 into rows 2 and 3
#undef CGIC_DPP_ADD
    return v;
}

// Block-wide exclusive scan of one value per thread (blockDim.x multiple of 64,
// <= 1024).  `smem` needs blockDim.x/64 + 1 elements.  Returns the exclusive
// prefix; *total receives the block total.  Contains three __syncthreads().
template <typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T *smem, T *total)
{
    const int lane = lane_id();
    const int wid = (int)(threadIdx.x >> 6);
    const int nw = (int)(blockDim.x >> 6);
    T inc = wave_inclusive_scan(v);
    if (lane == kWave - 1) smem[wid] = inc;
    __syncthreads();
    if (wid == 0) {
        T w = lane < nw ? smem[lane] : T(0);
        T winc = wave_inclusive_scan(w);
        if (lane < nw) smem[lane] = winc - w;  // exclusive offset of each wave
        if (lane == nw - 1) smem[nw] = winc;
    }
    __syncthreads();
    T res = smem[wid] + (inc - v);
    *total = smem[nw];
    __syncthreads();   // smem may be reused by the next call
    return res;
}
#endif

}  // namespace cgic
