// cgic_coder.hip -- the entropy-coder half of CGIC.compress:
//   encode: masked select + static-Huffman bit packing + 1-bit mask packing
//           (reference: CGIC/models/model.py:217-260, CGIC/tools/indices_coding.py:78-126,
//            CGIC/tools/mask_coding.py:14-55)
//   decode: prefix decoding, mask -> index scatter, x2/x4 merge, embedding gather
//           (reference: indices_coding.py:131-168, mask_coding.py:59-96, model.py:269-397)
//
// The reference does this per image on the host: .tolist() (device sync), Python
// string concatenation bit by bit, one file per stream.  Here every (image, stream)
// pair is one workgroup and nothing leaves the device:
//
//  encode  phase A: wave prefix sums (count | bit-length packed in one u64) compact the
//                   selected symbols and give each its end bit offset;
//          phase B: GATHER formulation -- one thread per 32-bit output word binary-searches
//                   the symbol that covers its first bit and ORs the overlapping code bits
//                   together.  No atomics, no pre-zeroed output, deterministic, and codes of
//                   any length (the reference's table can reach 1023 bits) need no special case.
//          masks:   64 positions per wave -> one __ballot -> 8 bit-reversed bytes.
//  decode  one wave per stream: every lane speculatively decodes "a codeword starting at
//          bit base+lane" through a 13-bit LUT in LDS (64 starts in one LDS round trip); a
//          wave-uniform scalar chain (v_readlane) then walks the true codeword boundaries,
//          ~10 cycles per symbol instead of a dependent LDS/HBM lookup per symbol.  Codes
//          longer than 13 bits fall back to a trie walk at the point the chain reaches them.
//  merge   rank = prefix popcount of the bit-packed masks (read straight from the mask
//          streams); one block per image scatters the three symbol lists, sums the x1/x2/x4
//          grids and gathers codebook rows.
// All integer / bit work: outputs are bit-identical to the reference by construction and
// checked against it through the oracle + tests/golden/{coders,compress_cfg1}.npz.
#include "cgic_common.h"

#include <atomic>

// VGPR caps of the per-image kernels (registers per lane).  What matters is not their own occupancy but what they
// leave to the kernels of OTHER batches in flight on the same CU (bench.py --lanes): a 512-thread VQ workgroup takes
// 2 x 152 of a SIMD's 512 registers per lane.
#ifndef CGIC_CAP_COMPRESS
#define CGIC_CAP_COMPRESS 48     // 49 uncapped, no spills at 48: 4 waves x 48 fit beside a VQ workgroup (86.1 -> 87.9 GPixel/s at 4 lanes)
#endif
#ifndef CGIC_CAP_DECODE
#define CGIC_CAP_DECODE 0        // 69 uncapped; 56 / 48 spill 13 / 36 registers and were measured slower (85.8 / 83.3)
#endif
#ifndef CGIC_CAP_MERGE
#define CGIC_CAP_MERGE 0
#endif
#if CGIC_CAP_COMPRESS
#define CGIC_VGPR_CAP_COMPRESS __attribute__((amdgpu_num_vgpr(CGIC_CAP_COMPRESS / 2)))
#else
#define CGIC_VGPR_CAP_COMPRESS
#endif
#if CGIC_CAP_DECODE
#define CGIC_VGPR_CAP_DECODE __attribute__((amdgpu_num_vgpr(CGIC_CAP_DECODE / 2)))
#else
#define CGIC_VGPR_CAP_DECODE
#endif
#if CGIC_CAP_MERGE
#define CGIC_VGPR_CAP_MERGE __attribute__((amdgpu_num_vgpr(CGIC_CAP_MERGE / 2)))
#else
#define CGIC_VGPR_CAP_MERGE
#endif

namespace cgic {

#ifdef CGIC_PHASE_CLOCKS
__device__ long long g_phase_clk[32];
__device__ long long g_blk_t[2 * 4096];
#endif

#ifndef CGIC_ENC_THREADS
#define CGIC_ENC_THREADS 1024
#endif
constexpr int kEncThreads = CGIC_ENC_THREADS;         // x kEncItems = 4096 positions per scan round: one round per 256x256 stream
constexpr int kEncItems = 4;            // consecutive positions per thread per scan round
constexpr int kLdsPos = 8192;           // streams up to this many positions keep phase-A results in LDS
constexpr int kDecLutMax = 1 << kLutBitsMax;        // 13-bit LUT

// -------------------------------------------------------------------------------------------
// encode
// -------------------------------------------------------------------------------------------
struct EncStorage {
    uint32_t *cend;   // inclusive end bit of each selected symbol
    uint16_t *csym;   // the symbol
};

// code bits [off, off+nb) of symbol s, right-aligned; 1 <= nb <= 32
__device__ __forceinline__ uint32_t code_bits(const TableDev &t, int s, uint32_t off, uint32_t nb)
{
    const uint32_t j = off >> 5;
    const uint32_t *c = t.code + (size_t)s * t.words;
    uint64_t w = (uint64_t)c[j] << 32;
    if ((int)(j + 1) < t.words) w |= c[j + 1];
    return (uint32_t)((w << (off & 31)) >> (64 - nb));
}

__device__ int pack_huffman_stream(const TableDev &t, unsigned long long carry, EncStorage st, uint8_t *out, int64_t cap);

// Phase A + B for one Huffman-coded stream.  sym_at(i, &flag) returns the symbol at linear
// position i and whether it is selected.  Returns bytes written (0 = empty file) or CGIC_ERR_*.
template <typename SymAt>
__device__ int encode_huffman_stream(const TableDev &t, int64_t npos, SymAt sym_at, EncStorage st,
                                     uint8_t *out, int64_t cap)
{
    __shared__ unsigned long long scan_smem[kEncThreads / kWave + 1];
    __shared__ int s_err;
    const int tid = threadIdx.x;
    if (tid == 0) s_err = 0;
    __syncthreads();
    CGIC_STAMP2(1);

    unsigned long long carry = 0;   // (count << 32) | bits, uniform
    for (int64_t base = 0; base < npos; base += (int64_t)kEncThreads * kEncItems) {
        int sym[kEncItems];
        uint32_t len[kEncItems];
        unsigned long long local = 0;
#pragma unroll
        for (int k = 0; k < kEncItems; ++k) {
            const int64_t i = base + (int64_t)tid * kEncItems + k;
            bool flag = false;
            int64_t s = 0;
            if (i < npos) s = sym_at(i, &flag);
            uint32_t l = 0;
            if (flag) {
                if (s < 0 || s >= t.n) { s_err = CGIC_ERR_INVALID; s = 0; }   // KeyError in the reference
                l = (uint32_t)t.len[s];
            }
            sym[k] = flag ? (int)s : -1;
            len[k] = l;
            local += flag ? ((1ull << 32) | l) : 0ull;
        }
        unsigned long long total;
        unsigned long long excl = block_exclusive_scan(local, scan_smem, &total) + carry;
#pragma unroll
        for (int k = 0; k < kEncItems; ++k) {
            if (sym[k] >= 0) {
                const uint32_t ci = (uint32_t)(excl >> 32);
                excl += (1ull << 32) | len[k];
                st.cend[ci] = (uint32_t)excl;
                st.csym[ci] = (uint16_t)sym[k];
            }
        }
        carry += total;
    }
    __syncthreads();   // phase-A stores (LDS or global, same workgroup) visible to phase B
    CGIC_STAMP2(2);
    if (s_err) return s_err;
    return pack_huffman_stream(t, carry, st, out, cap);
}

// Phase B: one thread per 32-bit output word gathers the code bits that overlap it (no atomics, any code length).
__device__ int pack_huffman_stream(const TableDev &t, unsigned long long carry, EncStorage st, uint8_t *out, int64_t cap)
{
    const int tid = threadIdx.x;
    const uint32_t count = (uint32_t)(carry >> 32);
    const uint32_t total_bits = (uint32_t)carry;
    if (count == 0) return 0;                                   // `if not text: write b''`  (:116-118)
    const uint32_t pad = 8 - (total_bits & 7);                  // 1..8                      (:92)
    const int64_t nbytes = 1 + (int64_t)((total_bits + pad) >> 3);
    if (nbytes > cap) return CGIC_ERR_CAPACITY;
    const int64_t nwords = (nbytes + 3) >> 2;
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);        // slot bases are 16-byte aligned
    for (int64_t q = tid; q < nwords; q += kEncThreads) {
        // stream bits [32q, 32q+32); stream bit = 8 (header byte) + payload bit
        const int64_t lo = 32 * q - 8, hi = lo + 32;
        uint32_t acc = q == 0 ? (pad << 24) : 0u;               // "{0:08b}".format(extra_padding) (:96)
        const uint32_t p0 = lo < 0 ? 0u : (uint32_t)lo;
        if (p0 < total_bits) {
            // first symbol whose end bit is > p0
            uint32_t a = 0, b = count;
            while (a < b) {
                uint32_t m = (a + b) >> 1;
                if (st.cend[m] > p0) b = m; else a = m + 1;
            }
            uint32_t c = a;
            uint32_t start = c ? st.cend[c - 1] : 0u;
            while (c < count && (int64_t)start < hi) {
                const uint32_t end = st.cend[c];
                const int s = st.csym[c];
                const uint32_t from = start > p0 ? start : p0;
                const uint32_t to = (int64_t)end < hi ? end : (uint32_t)hi;
                if (to > from) acc |= code_bits(t, s, from - start, to - from) << (uint32_t)(hi - to);
                start = end;
                ++c;
            }
        }
        out32[q] = __builtin_bswap32(acc);                       // MSB-first bytes (:108)
    }
    CGIC_STAMP2(3);
    return (int)nbytes;
}

// ---- split streams: the positions of a long stream are divided over several workgroups ("parts": contiguous position
// ranges).  A part compacts and sizes its own range, tells the others (symbols, bits, its first 32 code bits), learns where
// its bits start from the parts before it, and packs the output words whose FIRST payload bit is its own; the tail of its last
// word comes from the heads of the parts behind it.  No atomics on the output, no pre-zeroed buffer.
constexpr int kEncMaxParts = 7;           // descriptors of a stream fit two ticket slots: 4 words per part + the reader count
constexpr int kEncDoneWord = 4 * kEncMaxParts;
struct EncExchange {
    unsigned int *tick;      // 2 ticket slots (zero when the launch starts): [4g..4g+3] = {symbols, bits, head, 1 ok | 2 error}
    int part, nparts;
};

__device__ int pack_huffman_part(const TableDev &t, unsigned long long mine, EncStorage st, uint8_t *out, int64_t cap,
                                 const EncExchange &x, int err)
{
    __shared__ uint32_t s_desc[4 * kEncMaxParts];
    const int tid = threadIdx.x;
    const uint32_t count = err ? 0u : (uint32_t)(mine >> 32), bits = err ? 0u : (uint32_t)mine;
    if (tid == 0) {
        // the first (up to) 32 bits of this part's code bits, left-aligned
        uint32_t head = 0, pos = 0, start = 0;
        for (uint32_t c = 0; c < count && pos < 32; ++c) {
            const uint32_t end = st.cend[c], len = end - start;
            const uint32_t take = len < 32 - pos ? len : 32 - pos;
            head |= code_bits(t, st.csym[c], 0, take) << (32 - pos - take);
            pos += len;
            start = end;
        }
        // (agent-scope stores, drained, then the flag; agent-scope loads on the other side: no L2-wide fences, see part_exchange)
        unsigned int *d = x.tick + 4 * x.part;
        __hip_atomic_store(&d[0], count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&d[1], bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&d[2], head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(&d[3], err ? 2u : 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < x.nparts) {
        // every part reads every descriptor: the prefix needs the parts before, the last word and the header byte the ones behind
        unsigned int *d = x.tick + 4 * tid;
        unsigned int f;
        while ((f = __hip_atomic_load(&d[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        s_desc[4 * tid] = __hip_atomic_load(&d[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_desc[4 * tid + 1] = __hip_atomic_load(&d[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_desc[4 * tid + 2] = __hip_atomic_load(&d[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_desc[4 * tid + 3] = f;
    }
    __syncthreads();
    CGIC_STAMP2(4);
    if (tid == 0) {
        // the last part to have read zeroes the slots for the next launch
        const unsigned int old = __hip_atomic_fetch_add(&x.tick[kEncDoneWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned int)(x.nparts - 1)) {
            // (the data words too: the pool hands the slot to other kernels later, which expect every word zero)
            for (int g = 0; g < 4 * x.nparts; ++g) __hip_atomic_store(&x.tick[g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&x.tick[kEncDoneWord], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    uint32_t S = 0, total_bits = 0, total_count = 0;
    bool any_err = false;
    for (int g = 0; g < x.nparts; ++g) {
        if (g < x.part) S += s_desc[4 * g + 1];
        total_bits += s_desc[4 * g + 1];
        total_count += s_desc[4 * g];
        any_err |= s_desc[4 * g + 3] == 2u;
    }
    if (any_err) return CGIC_ERR_INVALID;                       // KeyError in the reference
    if (total_count == 0) return 0;                             // `if not text: write b''`  (:116-118)
    const uint32_t pad = 8 - (total_bits & 7);                  // 1..8                      (:92)
    const int64_t nbytes = 1 + (int64_t)((total_bits + pad) >> 3);
    if (nbytes > cap) return CGIC_ERR_CAPACITY;
    const int64_t nwords = (nbytes + 3) >> 2;
    const uint32_t E = S + bits;
    uint32_t *out32 = reinterpret_cast<uint32_t *>(out);
    // words whose first payload bit (0 for word 0, else 32 q - 8) lies in [S, E)
    int64_t q_lo = S == 0 ? 0 : ((int64_t)S + 8 + 31) >> 5, q_hi = bits ? ((int64_t)E + 8 + 31) >> 5 : q_lo;
    if (bits == 0) q_hi = q_lo = 0;
    for (int64_t q = q_lo + tid; q < q_hi; q += kEncThreads) {
        const int64_t lo = 32 * q - 8, hi = lo + 32;
        uint32_t acc = q == 0 ? (pad << 24) : 0u;               // "{0:08b}".format(extra_padding) (:96)
        const uint32_t p0 = (lo < 0 ? 0u : (uint32_t)lo) - S;   // relative to this part's first bit
        const int64_t rhi = hi - (int64_t)S;
        {
            uint32_t a = 0, b = count;
            while (a < b) {
                uint32_t m = (a + b) >> 1;
                if (st.cend[m] > p0) b = m; else a = m + 1;
            }
            uint32_t c = a;
            uint32_t start = c ? st.cend[c - 1] : 0u;
            while (c < count && (int64_t)start < rhi) {
                const uint32_t end = st.cend[c];
                const int sy = st.csym[c];
                const uint32_t from = start > p0 ? start : p0;
                const uint32_t to = (int64_t)end < rhi ? end : (uint32_t)rhi;
                if (to > from) acc |= code_bits(t, sy, from - start, to - from) << (uint32_t)(rhi - to);
                start = end;
                ++c;
            }
        }
        // the rest of the word: the first bits of the parts behind
        int64_t pos = E;
        for (int g = x.part + 1; g < x.nparts && pos < hi; ++g) {
            const uint32_t bg = s_desc[4 * g + 1];
            if (bg == 0) continue;
            const uint32_t have = bg < 32u ? bg : 32u, want = (uint32_t)(hi - pos);
            const uint32_t take = have < want ? have : want;
            acc |= (s_desc[4 * g + 2] >> (32 - take)) << (uint32_t)(hi - pos - take);
            pos += bg;
        }
        out32[q] = __builtin_bswap32(acc);                       // MSB-first bytes (:108)
    }
    if (x.part == x.nparts - 1) {
        // words that hold nothing but padding (first payload bit >= total_bits)
        for (int64_t q = (((int64_t)total_bits + 8 + 31) >> 5) + tid; q < nwords; q += kEncThreads) out32[q] = 0u;
    }
    CGIC_STAMP2(3);
    return (int)nbytes;
}

// Phase A for LONG streams (a 768x768 tile's fine grid has 36 864 positions): the round-by-round form above pays one
// block-wide scan (three barriers) per 4096 positions -- nine in a row for that stream, 22 us on one CU.  Here the
// (symbol | unselected) entries of ALL positions are first staged in LDS with coalesced loads (2 bytes each), every
// thread then owns one CONTIGUOUS chunk: it sums its code lengths, ONE block scan turns the sums into start offsets, and a
// second walk over the same LDS entries writes (end bit, symbol) of the selected ones.  The compacted list goes to the
// static LDS arrays when it fits (a fine stream keeps ~10 % of its positions), else to the global workspace.
// dense = non-NULL: the stream covers every position of a contiguous int64 index array with an int32 mask beside it (the fine
// grid): four positions per thread and trip, 16-byte loads, no index arithmetic
template <typename SymAt>
__device__ int encode_huffman_stream_long(const TableDev &t, int64_t npos, SymAt sym_at, uint16_t *stage, EncStorage st_lds,
                                          EncStorage st_glob, uint8_t *out, int64_t cap, const int64_t *dense_ind = nullptr,
                                          const int32_t *dense_mask = nullptr, const EncExchange *x = nullptr)
{
    __shared__ unsigned long long scan_smem[kEncThreads / kWave + 1];
    __shared__ int s_err;
    const int tid = threadIdx.x;
    if (tid == 0) s_err = 0;
    __syncthreads();
    // eight positions per thread in flight at a time: with one, every trip of the loop waited out its own HBM / L2 round
    // trip (36 trips x ~0.7 us for a 768x768 tile's fine stream)
    constexpr int kInFlight = 8;
    const bool dense = dense_ind != nullptr && (npos & 3) == 0 && ((reinterpret_cast<uintptr_t>(dense_ind) | reinterpret_cast<uintptr_t>(dense_mask)) & 15) == 0;
    if (dense) {
        const int64_t nq = npos >> 2;
        for (int64_t base = 0; base < nq; base += (int64_t)kEncThreads * 2) {
            int4 m[2];
            longlong2 a[2], b[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int64_t q = base + (int64_t)k * kEncThreads + tid;
                if (q < nq) {
                    m[k] = reinterpret_cast<const int4 *>(dense_mask)[q];
                    a[k] = reinterpret_cast<const longlong2 *>(dense_ind)[2 * q];
                    b[k] = reinterpret_cast<const longlong2 *>(dense_ind)[2 * q + 1];
                }
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int64_t q = base + (int64_t)k * kEncThreads + tid;
                if (q < nq) {
                    const int64_t v[4] = {a[k].x, a[k].y, b[k].x, b[k].y};
                    const int f[4] = {m[k].x, m[k].y, m[k].z, m[k].w};
                    uint32_t e[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        e[c] = 0xFFFFu;
                        if (f[c] == 1) {
                            if (v[c] < 0 || v[c] >= t.n || v[c] >= 0xFFFF) s_err = CGIC_ERR_INVALID;
                            else e[c] = (uint32_t)v[c];
                        }
                    }
                    reinterpret_cast<uint2 *>(stage)[q] = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
                }
            }
        }
    } else
    for (int64_t base = 0; base < npos; base += (int64_t)kEncThreads * kInFlight) {
        int64_t sy[kInFlight];
        bool fl[kInFlight];
#pragma unroll
        for (int k = 0; k < kInFlight; ++k) {
            const int64_t i = base + (int64_t)k * kEncThreads + tid;
            fl[k] = false;
            sy[k] = 0;
            if (i < npos) sy[k] = sym_at(i, &fl[k]);
        }
#pragma unroll
        for (int k = 0; k < kInFlight; ++k) {
            const int64_t i = base + (int64_t)k * kEncThreads + tid;
            uint16_t e = 0xFFFFu;
            if (fl[k]) {
                if (sy[k] < 0 || sy[k] >= t.n || sy[k] >= 0xFFFF) s_err = CGIC_ERR_INVALID;      // KeyError in the reference
                else e = (uint16_t)sy[k];
            }
            if (i < npos) stage[i] = e;
        }
    }
    __syncthreads();
    CGIC_STAMP2(1);
    if (s_err) return x ? pack_huffman_part(t, 0ull, st_lds, out, cap, *x, s_err) : s_err;      // (the other parts wait for this one's word)
    // chunks of a multiple of 4 entries (8-byte LDS reads; `stage` is 16-byte aligned and padded by the caller's sizing)
    const int64_t chunk = ((npos + kEncThreads - 1) / kEncThreads + 3) & ~(int64_t)3;
    const int64_t lo = (int64_t)tid * chunk < npos ? (int64_t)tid * chunk : npos;
    const int64_t hi = lo + chunk < npos ? lo + chunk : npos;
    uint32_t lcount = 0, lbits = 0;
    for (int64_t i = lo; i < hi; i += 4) {
        const uint2 q = *reinterpret_cast<const uint2 *>(stage + i);       // (entries past npos read stale LDS: masked below)
        const uint32_t e[4] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16};
        uint32_t l[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) l[c] = (uint32_t)t.len[e[c] == 0xFFFFu ? 0 : e[c]];     // four lookups in flight
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const bool on = e[c] != 0xFFFFu && i + c < hi;
            lcount += on ? 1u : 0u;
            lbits += on ? l[c] : 0u;
        }
    }
    unsigned long long local = ((unsigned long long)lcount << 32) | lbits;
    unsigned long long total;
    unsigned long long excl = block_exclusive_scan(local, scan_smem, &total);
    const uint32_t count = (uint32_t)(total >> 32);
    const EncStorage st = count <= (uint32_t)kLdsPos ? st_lds : st_glob;
    uint32_t ci = (uint32_t)(excl >> 32), bit = (uint32_t)excl;
    for (int64_t i = lo; i < hi; i += 4) {
        const uint2 q = *reinterpret_cast<const uint2 *>(stage + i);
        const uint32_t e[4] = {q.x & 0xFFFFu, q.x >> 16, q.y & 0xFFFFu, q.y >> 16};
        uint32_t l[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) l[c] = (uint32_t)t.len[e[c] == 0xFFFFu ? 0 : e[c]];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (e[c] != 0xFFFFu && i + c < hi) {
                bit += l[c];
                st.cend[ci] = bit;
                st.csym[ci] = (uint16_t)e[c];
                ++ci;
            }
        }
    }
    __syncthreads();
    CGIC_STAMP2(2);
    if (x) return pack_huffman_part(t, total, st, out, cap, *x, 0);
    return pack_huffman_stream(t, total, st, out, cap);
}

// 1-bit stream (BinaryCoding): bit_at(i) in {0,1}; other values are a KeyError in the reference
template <typename BitAt>
__device__ int encode_binary_stream(int64_t npos, BitAt bit_at, uint8_t *out, int64_t cap)
{
    if (npos == 0) return 0;
    const uint32_t pad = 8 - (uint32_t)(npos & 7);
    const int64_t payload = (npos >> 3) + 1;
    const int64_t nbytes = 1 + payload;
    if (nbytes > cap) return CGIC_ERR_CAPACITY;
    __shared__ int b_err;
    const int tid = threadIdx.x, lane = lane_id();
    if (tid == 0) { b_err = 0; out[0] = (uint8_t)pad; }
    __syncthreads();
    const int64_t rounded = (npos + 8 + 63) & ~(int64_t)63;     // cover the pad byte too
    for (int64_t i = tid; i < rounded; i += kEncThreads) {
        int v = 0;
        if (i < npos) {
            v = bit_at(i);
            if (v != 0 && v != 1) { b_err = CGIC_ERR_INVALID; v = 0; }
        }
        const unsigned long long bal = __ballot(v == 1);
        if (lane < 8) {
            const int64_t byte = ((i - lane) >> 3) + lane;
            if (byte < payload) out[1 + byte] = (uint8_t)(__brev((uint32_t)((bal >> (8 * lane)) & 0xFF)) >> 24);
        }
    }
    __syncthreads();
    return b_err ? b_err : (int)nbytes;
}

struct CompressArgs {
    TableDev tab;
    const int64_t *ind;
    const int32_t *mc, *mm, *mf;
    int64_t h, w;
    int stream_mask;        // bit s set = stream s written in this mode
    uint8_t *out;
    int64_t slot;
    int32_t *nbytes;        // [B, 5]
    uint32_t *ws_end;       // global phase-A storage for streams > kLdsPos positions
    uint16_t *ws_sym;
    int64_t ws_stride;      // positions reserved per (image, stream) in the workspace
    unsigned long long *hist;   // optional [tab.n]: usage histogram of ALL h*w indices (job 5 of each image)
    int64_t stage_positions;    // entries of the dynamic-LDS staging buffer (0: none)
    int parts[3];               // workgroups per index stream (coarse, medium, fine); > 1: split streams, see EncExchange
    unsigned int *tick;         // [B, 3] x 2 ticket slots when any stream is split
};

constexpr int kLdsTable = 1024;          // tables up to this many single-word codes are staged in LDS

__global__ __launch_bounds__(kEncThreads) CGIC_VGPR_CAP_COMPRESS void compress_streams_kernel(CompressArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_end[kLdsPos];
    __shared__ uint16_t lds_sym[kLdsPos];
    __shared__ int32_t lds_len[kLdsTable];
    __shared__ uint32_t lds_code[kLdsTable];
    extern __shared__ __attribute__((aligned(16))) uint16_t lds_stage[];     // [h*w] for long streams (see encode_huffman_stream_long)
    // grid (B, jobs): workgroups are dispatched image-fastest, the LONG jobs first (fine, medium, coarse indices,
    // then the two mask streams, then the histogram).  1024-thread workgroups are handed out at ~100 per us: with the
    // job as the fast index the fine stream of the last image started 5 us late and ended the launch.
    // (split streams: the parts of the fine stream, then the medium one's, ...)
    // Split streams wait for each other: their launch is (jobs, B) -- the parts of a stream are neighbours in dispatch order,
    // so a workgroup never holds a CU waiting for one that is hundreds of workgroups behind it in the queue.
    const bool jobs_fastest = a.tick != nullptr;
    int s, part = 0, nparts = 1;
    {
        int y = (int)(jobs_fastest ? blockIdx.x : blockIdx.y);
        if (y < a.parts[2]) { s = 2; part = y; nparts = a.parts[2]; }
        else if ((y -= a.parts[2]) < a.parts[1]) { s = 1; part = y; nparts = a.parts[1]; }
        else if ((y -= a.parts[1]) < a.parts[0]) { s = 0; part = y; nparts = a.parts[0]; }
        else { y -= a.parts[0]; s = y == 0 ? 4 : y == 1 ? 3 : 5; }
    }
    const int64_t b = jobs_fastest ? blockIdx.y : blockIdx.x;
    CGIC_STAMP2(0);
    CGIC_SPAN_BEGIN();
#ifdef CGIC_PHASE_CLOCKS      // dev: per-workgroup (start, end) for tools/probe_compress_blocks.py
    const unsigned int dbg_lin = (unsigned int)b * 16 + (jobs_fastest ? blockIdx.x : blockIdx.y);      // (image, job in launch order) like the probe expects
    if (threadIdx.x == 0 && dbg_lin < 4096) g_blk_t[2 * dbg_lin] = wall_clock64();
    struct DbgEnd { unsigned int lin; __device__ ~DbgEnd() { if (threadIdx.x == 0 && lin < 4096) g_blk_t[2 * lin + 1] = wall_clock64(); } } dbg_end{dbg_lin};
#endif
    if (s == CGIC_NUM_STREAMS) {
        // job 5: usage histogram of this image's indices (quantize.py:79-81) -- LDS histogram, then
        // at most one global atomic per non-empty bin per image
        unsigned int *lh = lds_end;
        const int K = a.tab.n;
        for (int k = threadIdx.x; k < K; k += kEncThreads) lh[k] = 0;
        __syncthreads();
        const int64_t n = a.h * a.w;
        const int64_t *ind = a.ind + b * n;
        for (int64_t i = threadIdx.x; i < n; i += kEncThreads) {
            const int64_t v = ind[i];
            if (v >= 0 && v < K) atomicAdd(&lh[v], 1u);
        }
        __syncthreads();
        for (int k = threadIdx.x; k < K; k += kEncThreads)
            if (lh[k]) atomicAdd(&a.hist[k], (unsigned long long)lh[k]);
        return;
    }
    int32_t *nb = a.nbytes + b * CGIC_NUM_STREAMS + s;
    if (!((a.stream_mask >> s) & 1)) {
        if (threadIdx.x == 0 && part == nparts - 1) *nb = -1;
        return;
    }
    uint8_t *out = a.out + (b * CGIC_NUM_STREAMS + s) * a.slot;
    const int64_t h = a.h, w = a.w;
    int rc;
    if (s < 3) {
        // code table -> LDS (length + code lookups then cost an LDS access, not an L2 round trip each)
        if (a.tab.n <= kLdsTable && a.tab.words == 1) {
            for (int i = threadIdx.x; i < a.tab.n; i += kEncThreads) { lds_len[i] = a.tab.len[i]; lds_code[i] = a.tab.code[i]; }
            a.tab.len = lds_len;
            a.tab.code = lds_code;
            __syncthreads();
        }
        const int sh = 2 - s;                                   // stride 4, 2, 1
        const int64_t gh = h >> sh, gw = w >> sh, npos = gh * gw;
        const int32_t *mask = (s == 0 ? a.mc : s == 1 ? a.mm : a.mf) + b * npos;
        const int64_t *ind = a.ind + b * h * w;
        // this workgroup's positions: all of them, or the part-th of nparts ranges (whole groups of four)
        const int64_t per = nparts > 1 ? (((npos + nparts - 1) / nparts) + 3) & ~(int64_t)3 : npos;
        const int64_t pos0 = (int64_t)part * per < npos ? (int64_t)part * per : npos;
        const int64_t mypos = pos0 + per < npos ? per : npos - pos0;
        EncStorage st, st_glob;
        st.cend = lds_end; st.csym = lds_sym;
        st_glob.cend = a.ws_end + (b * 3 + s) * a.ws_stride + pos0;
        st_glob.csym = a.ws_sym + (b * 3 + s) * a.ws_stride + pos0;
        // ind[:, ::4, ::4][mask_c == 1] etc.: row-major over the granularity's own grid (:219-221)
        auto sym_at = [&](int64_t i, bool *flag) -> int64_t {
            // both loads are issued unconditionally so that they share one memory round trip
            const int ii = (int)(i + pos0), gwi = (int)gw;              // 32-bit divide (h*w < 2^26)
            const int y = ii / gwi, x = ii - y * gwi;
            const int64_t v = ind[((int64_t)(y << sh) * w) + (x << sh)];
            *flag = mask[ii] == 1;
            return v;
        };
        if (nparts > 1) {
            const EncExchange ex{a.tick + ((b * 3 + s) * 2) * kTicketStride, part, nparts};
            rc = encode_huffman_stream_long(a.tab, mypos, sym_at, lds_stage, st, st_glob, out, a.slot, sh == 0 ? ind + pos0 : nullptr,
                                            sh == 0 ? mask + pos0 : nullptr, &ex);
        } else if (npos <= kLdsPos) rc = encode_huffman_stream(a.tab, npos, sym_at, st, out, a.slot);
        else if (a.stage_positions >= npos)
            rc = encode_huffman_stream_long(a.tab, npos, sym_at, lds_stage, st, st_glob, out, a.slot, sh == 0 ? ind : nullptr, sh == 0 ? mask : nullptr);
        else rc = encode_huffman_stream(a.tab, npos, sym_at, st_glob, out, a.slot);
    } else {
        const int sh = s == 3 ? 2 : 1;
        const int64_t npos = (h >> sh) * (w >> sh);
        const int32_t *mask = (s == 3 ? a.mc : a.mm) + b * npos;   // grain_mask[k].flatten() (:230-231)
        rc = encode_binary_stream(npos, [&](int64_t i) { return (int)mask[i]; }, out, a.slot);
    }
    if (threadIdx.x == 0 && part == nparts - 1) *nb = rc < 0 ? rc - 10 : rc;     // errors are CGIC_ERR_* - 10 (-1 means "not written")
    CGIC_SPAN_END();
}

struct EncodeOneArgs {
    TableDev tab;
    const void *syms;
    int elem_bytes;
    int64_t n;
    uint8_t *out;
    int64_t cap;
    int32_t *nbytes;
    uint32_t *ws_end;
    uint16_t *ws_sym;
};

__global__ __launch_bounds__(kEncThreads) void encode_stream_kernel(EncodeOneArgs a)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_end[kLdsPos];
    __shared__ uint16_t lds_sym[kLdsPos];
    EncStorage st;
    if (a.n <= kLdsPos) { st.cend = lds_end; st.csym = lds_sym; }
    else { st.cend = a.ws_end; st.csym = a.ws_sym; }
    auto sym_at = [&](int64_t i, bool *flag) -> int64_t {
        *flag = true;
        return a.elem_bytes == 8 ? reinterpret_cast<const int64_t *>(a.syms)[i]
                                 : (int64_t) reinterpret_cast<const int32_t *>(a.syms)[i];
    };
    int rc;
    if (a.tab.n == 2 && a.tab.max_len == 1) {
        // BinaryCoding's fixed table: ballot path (symbol v <-> bit v)
        rc = encode_binary_stream(a.n, [&](int64_t i) { bool f; return (int)sym_at(i, &f); }, a.out, a.cap);
    } else {
        rc = encode_huffman_stream(a.tab, a.n, sym_at, st, a.out, a.cap);
    }
    if (threadIdx.x == 0) *a.nbytes = rc;
}

// -------------------------------------------------------------------------------------------
// decode
// -------------------------------------------------------------------------------------------
constexpr int kWinBytes = 8192;          // LDS window of stream bytes per decoding wave
constexpr int kWinWords = kWinBytes / 4 + 4;

// One wave decodes one stream.  The stream is staged through an LDS window (coalesced 16-byte
// loads), every lane looks up "the codeword starting at bit base+lane" in the LDS LUT one chunk
// AHEAD of the scalar chain that walks the true boundaries, so the chain (v_readlane + SALU,
// wave-uniform) is the only serial part.  put(k, sym) stores the k-th symbol.
// Returns the symbol count, -1 for an empty input (None in the reference, :158-159).
struct WaveDecoder {
    const TableDev &t;
    const uint32_t *lut;     // LDS
    uint32_t *win;           // LDS, kWinWords
    const uint8_t *in;       // global; in[0] is the pad-count byte
    int nbytes;
    int wb;                  // first stream byte held in the window (multiple of 4)

    __device__ __forceinline__ void fill(int first_byte)
    {
        // window = stream bytes [wb, wb + kWinBytes + 16), wb 4-aligned relative to the (16-byte
        // aligned or not) base pointer: use aligned dword loads of the global buffer
        const int lane = lane_id();
        wb = first_byte & ~3;
        const uintptr_t g = reinterpret_cast<uintptr_t>(in) + (uintptr_t)wb;
        const uint32_t *ga = reinterpret_cast<const uint32_t *>(g & ~(uintptr_t)3);
        wsh = (int)(g & 3);          // the window is shifted by this many bytes w.r.t. wb
        const int limit = (nbytes - wb + wsh + 3) / 4 + 2;   // dwords that may be touched (slack in the buffer contract)
        for (int k = lane; k < kWinWords; k += kWave) win[k] = k < limit ? ga[k] : 0u;
        __builtin_amdgcn_wave_barrier();
    }
    int wsh;

    // 32 payload bits starting at payload bit p (MSB first); caller guarantees the window covers them
    __device__ __forceinline__ uint32_t fetch32(int p) const
    {
        const int o = 1 + (p >> 3) - wb + wsh;       // byte offset inside the window
        const uint32_t a = win[o >> 2], b = win[(o >> 2) + 1];
        const uint64_t w = ((uint64_t)__builtin_bswap32(a) << 32) | __builtin_bswap32(b);
        return (uint32_t)((w << (8 * (o & 3) + (p & 7))) >> 32);
    }
    __device__ __forceinline__ bool covers(int p_last) const
    {   // bytes needed: up to stream byte 1 + (p_last >> 3) + 4 (+ shift)
        return 1 + (p_last >> 3) + 8 + wsh < wb + kWinBytes;
    }

    template <typename Put>
    __device__ int run(int cap, Put put, int *overflow)
    {
        if (nbytes <= 0) return -1;
        const int lane = lane_id();
        fill(0);
        const int pad = (int)(__builtin_bswap32(win[wsh >> 2]) >> (24 - 8 * (wsh & 3))) & 0xFF;   // remove_padding :131-138
        const int total = (nbytes - 1) * 8;
        int nbits = pad == 0 ? 0 : total - pad;                  // text[:-0] is empty in Python
        if (nbits < 0) nbits = 0;
        const int LB = t.lut_bits;
        int pos = 0, count = 0;
        bool done = false;
        // prologue: lookups for chunk 0
        uint32_t e_cur = 0xFFFFFF00u;
        if (lane < nbits) e_cur = lut[fetch32(lane) >> (32 - LB)];
        for (int base = 0; base < nbits && !done; base += kWave) {
            // lookups for the NEXT chunk, issued before this chunk's chain
            const int nbase = base + kWave;
            uint32_t e_next = 0xFFFFFF00u;
            if (nbase < nbits) {
                if (!covers(nbase + kWave)) fill(1 + (nbase >> 3));
                const int p = nbase + lane;
                if (p < nbits) e_next = lut[fetch32(p) >> (32 - LB)];
            }
            const int L = (int)(e_cur & 0xFF);
            int S = (int)(e_cur >> 8);
            unsigned long long starts = 0;
            while (pos < base + kWave) {
                const int i = pos - base;
                int Li = __builtin_amdgcn_readlane(L, i);
                const int Si = __builtin_amdgcn_readlane(S, i);
                if (Li == 0) {
                    // code longer than the LUT window: continue in the trie from node Si
                    if (Si == 0xFFFFFF) { done = true; break; }
                    int node = Si, sym = -1;
                    int q = pos + LB;
                    while (q < nbits) {
                        const int bit = (in[1 + (q >> 3)] >> (7 - (q & 7))) & 1;
                        const int c = t.child[2 * node + bit];
                        ++q;
                        if (c == INT32_MIN) break;
                        if (c < 0) { sym = ~c; break; }
                        node = c;
                    }
                    if (sym < 0) { done = true; break; }         // out of bits: trailing partial code is dropped
                    Li = q - pos;
                    S = lane == i ? sym : S;
                }
                if (pos + Li > nbits) { done = true; break; }
                starts |= 1ull << i;
                pos += Li;
            }
            if (starts) {
                const int rank = __popcll(starts & ((1ull << lane) - 1ull));
                if ((starts >> lane) & 1ull) {
                    if (count + rank < cap) put(count + rank, S);
                    else *overflow = 1;
                }
                count += __popcll(starts);
            }
            e_cur = e_next;
        }
        return count;
    }
};

__device__ __forceinline__ void load_lut(const TableDev &t, uint32_t *lut)
{
    const int n = 1 << t.lut_bits;
    for (int i = threadIdx.x; i < n; i += blockDim.x) lut[i] = t.lut[i];
}

struct DecodeOneArgs {
    TableDev tab;
    const uint8_t *in;
    int64_t nbytes;
    int64_t *syms;
    int64_t cap;
    int64_t *count;
};

// -------------------------------------------------------------------------------------------
// Parallel prefix-code decoding inside ONE stream (fast mode, max code length <= 64 bits).
//
// A stream is cut into 64-bit chunks; a wave owns a contiguous range of chunks.  Where the
// first codeword of a chunk starts depends on everything before it, so each wave first builds
// the FUNCTION  entry offset e in [0,64)  ->  (exit offset into the chunk after its range,
// number of symbols decoded)  without knowing e:
//   lane i looks up the codeword starting at bit i of the chunk (LUT in LDS, long codes by a
//   per-lane trie walk) -> next[i] = i + len, cnt[i] = 1;  six rounds of pointer doubling with
//   ds_bpermute turn next/cnt into "first position >= 64 reached from i / symbols on the way";
//   the chunk function is folded into the wave's running function with two more bpermutes.
// Functions of consecutive waves are composed through LDS (<= 16 scalar steps), which gives
// every wave its true entry offset and output index; then all waves decode their ranges
// concurrently with the scalar chain of WaveDecoder::run.  Exact for every table with
// max_len <= 64; longer tables (all-zero frequency counters give 224-bit codes) take the
// single-wave path.  A codeword never spans more than two chunks in fast mode, so every
// entry offset is < 64.
// -------------------------------------------------------------------------------------------
constexpr int kDecThreads = 1024;
constexpr int kDecWaves = kDecThreads / kWave;
constexpr int kSegWin = 1024;                      // LDS window of stream bytes per wave
constexpr int kSegWinWords = kSegWin / 4 + 4;           // 65 x 16 B: one uint4 per lane + one tail
constexpr int kBig = 1 << 28;                      // "past the end of the stream"
constexpr int kU = 10;                             // chunks in flight per wave (a 256x256 medium stream is ~150 chunks = 10 per wave: one round)
constexpr int kLdsTrieNodes = 2048;                // decode tries up to this many nodes are staged in LDS (16 KB)
constexpr int kFastChunks = 192;                   // streams up to this many chunks (1.5 KB) cache per-position
                                                   // lengths / symbols / chunk functions for the lane-per-chunk pass C
constexpr int kPackBig = 0xFF;                     // packed "past the end" marker (max real next = 63 + 64)
constexpr int kDecParts = 8;                        // workgroups per stream in the two-launch split form (decode_functions / decode_parts)
#ifndef CGIC_DEC_PARTS_MAX
#define CGIC_DEC_PARTS_MAX 12
#endif
constexpr int kDecPartsMax = CGIC_DEC_PARTS_MAX;                    // ... and at most in the one-launch form: flags 0..11 and the reader count (word 15) share one ticket slot
constexpr int kDecDoneWord = 15;
#ifndef CGIC_DEC_PART_BYTES
#define CGIC_DEC_PART_BYTES 1280
#endif
constexpr int kDecPartBytes = CGIC_DEC_PART_BYTES;     // stream bytes per part: 160 chunks, one pass-A round of 16 waves x 10 chunks

struct BitWindow {
    uint32_t *win;           // LDS, kSegWinWords
    const uint8_t *in;       // global; in[0] is the pad-count byte
    int nbytes;
    int wb, wsh;

    __device__ __forceinline__ void fill(int first_byte)
    {
        // one 16-byte load per lane (+1 tail) -> a single global round trip per refill
        const int lane = lane_id();
        wb = first_byte & ~3;
        const uintptr_t g = reinterpret_cast<uintptr_t>(in) + (uintptr_t)wb;
        const uint4 *ga = reinterpret_cast<const uint4 *>(g & ~(uintptr_t)15);
        wsh = (int)(g & 15);         // the window starts this many bytes before stream byte wb
        const int limit = (nbytes - wb + wsh + 15) / 16 + 1;      // 16-byte words that may be touched
        uint4 v0 = {0u, 0u, 0u, 0u}, v1 = {0u, 0u, 0u, 0u};
        if (lane < limit) v0 = ga[lane];
        if (lane == 0 && kWave < limit) v1 = ga[kWave];
        reinterpret_cast<uint4 *>(win)[lane] = v0;
        if (lane == 0) reinterpret_cast<uint4 *>(win)[kWave] = v1;
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ bool covers(int p_last) const { return 1 + (p_last >> 3) + 8 + wsh < wb + kSegWin - 16; }
    __device__ __forceinline__ uint32_t fetch32(int p) const
    {
        const int o = 1 + (p >> 3) - wb + wsh;
        const uint32_t a = __builtin_bswap32(win[o >> 2]), b = __builtin_bswap32(win[(o >> 2) + 1]);
        const uint32_t sh = (uint32_t)(8 * (o & 3) + (p & 7));          // 0..31 bits into the big-endian pair
        // (a << sh) | (b >> (32 - sh)) as ONE v_alignbit_b32 (a 64-bit vector shift is several times dearer);
        // alignbit's shift is mod 32, so sh == 0 needs the select
        const uint32_t r = __builtin_amdgcn_alignbit(a, b, 32u - sh);
        return sh ? r : a;
    }
    __device__ __forceinline__ int bit(int p) const
    {
        const int o = 1 + (p >> 3) - wb + wsh;
        return (int)((win[o >> 2] >> (8 * (o & 3) + 7 - (p & 7))) & 1u);
    }
    // the same as fetch32 with the address split: consecutive 64-bit chunks of one lane differ by
    // exactly two words and keep the same shift, so a chunk loop only adds 2 to `wi`
    __device__ __forceinline__ void locate(int p, int *wi, uint32_t *sh) const
    {
        const int o = 1 + (p >> 3) - wb + wsh;
        *wi = o >> 2;
        *sh = (uint32_t)(8 * (o & 3) + (p & 7));
    }
    __device__ __forceinline__ uint32_t fetch32_at(int wi, uint32_t sh) const
    {
        const uint32_t a = __builtin_bswap32(win[wi]), b = __builtin_bswap32(win[wi + 1]);
        const uint32_t r = __builtin_amdgcn_alignbit(a, b, 32u - sh);
        return sh ? r : a;
    }
};

// codeword starting at payload bit p: returns its length (0 = no complete codeword before
// nbits) and symbol.  Per-lane; long codes walk the trie (window must cover p + 64 + 32 bits).
// `bits` = the 32 payload bits starting at p (BitWindow::fetch32(p)).
__device__ __forceinline__ int codeword_at(const TableDev &t, const uint32_t *lut, const BitWindow &bw,
                                           int p, int nbits, int *sym, uint32_t bits)
{
    if (p >= nbits) return 0;
    const uint32_t e = lut[bits >> (32 - t.lut_bits)];
    int L = (int)(e & 0xFF);
    int S = (int)(e >> 8);
    if (L == 0) {
        if (S == 0xFFFFFF) return 0;
        int node = S, q = p + t.lut_bits;
        S = -1;
        while (q < nbits) {
            const int c = t.child[2 * node + bw.bit(q)];
            ++q;
            if (c == INT32_MIN) break;
            if (c < 0) { S = ~c; break; }
            node = c;
        }
        if (S < 0) return 0;
        L = q - p;
    }
    if (p + L > nbits) return 0;       // trailing partial codeword: dropped by the reference
    *sym = S;
    return L;
}
__device__ __forceinline__ int codeword_at(const TableDev &t, const uint32_t *lut, const BitWindow &bw,
                                           int p, int nbits, int *sym)
{
    return p < nbits ? codeword_at(t, lut, bw, p, nbits, sym, bw.fetch32(p)) : 0;
}

struct SegShared {
    int F[kDecWaves][kWave];          // exit offset of wave's range as a function of entry offset
    int C[kDecWaves][kWave];          // symbols decoded as a function of entry offset
};
struct FastTables {                   // per chunk x bit offset, filled by pass A when the stream is small enough
    uint8_t len[kFastChunks * kWave];     // codeword length starting there (0 = none)
    uint16_t sym[kFastChunks * kWave];    // its symbol
    uint16_t fn[kFastChunks * kWave];     // chunk function: low byte exit offset + 64 (0xFF = end), high byte symbols
};

// ---- split streams in ONE launch: the workgroups of a stream exchange their range functions through global memory
// (decode_split_kernel).  tick: one zeroed ticket slot per stream -- words 0..parts-2 "function of part g published",
// word 15 = parts that have read their predecessors; the last reader zeroes the slot again for the next launch.
struct PartSync {
    uint32_t *bf;            // [parts][64] range functions of this stream (global)
    unsigned int *tick;      // [kTicketStride] ticket slot of this stream (global, zero when the launch starts)
    int *s_entry;            // LDS [2]: the range's true entry offset and the symbols before it
};

// Wave 0 of a part: publish the range's function (`fn` = lane-th entry) and compose the functions of the parts before.
// A part only ever waits for parts with smaller workgroup ids, which were dispatched before it.
__device__ __forceinline__ void part_exchange(int part, int nparts, uint32_t fn, const PartSync &ps)
{
    const int lane = lane_id();
    // Hand-off without fences: the 64 words go out as agent-scope (write-through) stores, are drained, then the flag; the
    // readers use agent-scope loads.  An agent-scope release / acquire pair writes back and invalidates the XCD's whole L2
    // -- measured +4.6 us on the decode launch of 64 256x256 images with ONE stream split in two.
    if (part < nparts - 1) {                                     // nobody reads the last range's function
        __hip_atomic_store(&ps.bf[part * kWave + lane], fn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(&ps.tick[part], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    int e = 0, n = 0;
    if (part > 0) {
        if (lane == 0)
            for (int g = 0; g < part; ++g)
                while (__hip_atomic_load(&ps.tick[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 1u) __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        uint32_t r[kDecPartsMax - 1];
#pragma unroll
        for (int g = 0; g < kDecPartsMax - 1; ++g)
            r[g] = g < part ? __hip_atomic_load(&ps.bf[g * kWave + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
        for (int g = 0; g < kDecPartsMax - 1; ++g) {
            if (g < part && e < kWave) {
                const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)r[g], __builtin_amdgcn_readfirstlane(e));
                n += (int)(v >> 8);
                e = (v & 0xFF) == 0xFF ? kBig : (int)(v & 0xFF);
            }
        }
        if (lane == 0) {
            const unsigned int old = __hip_atomic_fetch_add(&ps.tick[kDecDoneWord], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old == (unsigned int)(nparts - 2)) {             // every part after the first has read: reset for the next launch
                for (int i = 0; i < nparts - 1; ++i) __hip_atomic_store(&ps.tick[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(&ps.tick[kDecDoneWord], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (lane == 0) { ps.s_entry[0] = e; ps.s_entry[1] = n; }
}

// Decode stream bytes `in` with waves [w0, w0+nw) of the block; each participating wave calls
// this with k = its index inside the stream.  Two block-wide barriers inside (ALL waves of the
// block must reach them, also waves with nw == 0 work: pass nw=0 and they just sync).
template <bool MULTI = true, typename Put>
__device__ __forceinline__ void decode_segmented(const TableDev &t, const uint32_t *lut, uint32_t *win,
                                                 SegShared *sh, const uint8_t *in, int nbytes, int pad, int w0, int nw,
                                                 int k, int cap, Put put, int *count_out, FastTables *ft = nullptr,
                                                 int part = 0, int nparts = 1, int e_in0 = 0, int n_in0 = 0,
                                                 uint32_t *bf_out = nullptr, const PartSync *sync = nullptr)
{
    // part / nparts: this workgroup handles the part-th of nparts equal chunk ranges of the stream (split streams,
    // see decode_functions_kernel); e_in0 / n_in0: bit offset into the range's first chunk where its first codeword
    // starts and the symbols before it; bf_out != NULL: only build the range's function (entry offset -> exit
    // offset | symbols << 8, 0xFF = past the end) and store its 64 entries there -- no symbols are written.
    const int lane = lane_id();
    const int wave = threadIdx.x >> 6;
    int nbits = 0, nchunks = 0, c0 = 0, c1 = 0;
    BitWindow bw{win, in, nbytes, 0, 0};
    const bool active = nw > 0 && nbytes > 0;
    if (nbytes > 0) {                                            // (every wave, also the idle ones: the segment loop below is uniform)
        nbits = pad == 0 ? 0 : (nbytes - 1) * 8 - pad;           // remove_padding :131-138
        if (nbits < 0) nbits = 0;
        nchunks = (nbits + kWave - 1) / kWave;
    }
    // Streams longer than the per-position tables (192 chunks = 1.5 KB) are decoded SEGMENT by segment of 160
    // chunks, each with the full three passes and the lane-per-chunk final pass; the exit (offset, count) of one
    // segment is the entry of the next.  (Before: one pass A over everything and a scalar chain per chunk --
    // 128 us for the streams of a 768x768 tile.)  nchunks is per stream, so every wave agrees on the loop; a
    // stream that long keeps all the workgroup's waves busy (the callers give one wave per 16 bytes), so every
    // wave also computes the same exit.
    const bool tables = ft != nullptr;
    constexpr int kSegChunks = kDecWaves * kU;                  // 160: one pass-A round per wave and segment
    static_assert(kSegChunks <= kFastChunks, "segment must fit the per-position tables");
    const int per_part = (nchunks + nparts - 1) / nparts;
    const int r_lo = part * per_part < nchunks ? part * per_part : nchunks;
    const int r_hi = r_lo + per_part < nchunks ? r_lo + per_part : nchunks;
    const int nseg = MULTI && tables && r_hi - r_lo > kFastChunks ? (r_hi - r_lo + kSegChunks - 1) / kSegChunks : 1;     // MULTI = false: the caller knows
    int e_in = e_in0, n_in = n_in0;
    for (int sg = 0; sg < nseg; ++sg) {
    const int seg_lo = nseg > 1 ? r_lo + sg * kSegChunks : r_lo;
    const int seg_hi = nseg > 1 ? (seg_lo + kSegChunks < r_hi ? seg_lo + kSegChunks : r_hi) : r_hi;
    if (active) {
        c0 = seg_lo + (int)((int64_t)k * (seg_hi - seg_lo) / nw);
        c1 = seg_lo + (int)((int64_t)(k + 1) * (seg_hi - seg_lo) / nw);
    }
    const bool fast = tables && seg_hi - seg_lo <= kFastChunks;      // wave-uniform
    FastTables *ftb = ft;                                            // tables are indexed from the segment's first chunk
    const int fo = seg_lo * kWave;
    // ---- pass A: range function (F, C) by pointer doubling; kU chunks in flight per wave so that
    // the LDS round trips of independent chunks overlap (one wave per SIMD has no other cover)
    CGIC_STAMP3(2);
    int F = lane, C = 0;
    if (active && c1 > c0) {
        bw.fill(1 + ((c0 * kWave) >> 3));
        if (c0 == 0) CGIC_STAMP3(16);
        for (int c = c0; c < c1; c += kU) {
            if (!bw.covers((c + kU + 2) * kWave)) bw.fill(1 + ((c * kWave) >> 3));
            // (next, count) packed in one word -> ONE ds_bpermute per doubling round (the LDS crossbar is
            // what bounds this pass): bits 0..7 = next position (0..127, kPackBig = past the stream),
            // bits 8..15 = codewords on the way
            int pk[kU];
            int wi0;
            uint32_t sh0;
            bw.locate(c * kWave + lane, &wi0, &sh0);
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                int sym = 0;
                const int L = c + u < c1 ? codeword_at(t, lut, bw, (c + u) * kWave + lane, nbits, &sym, bw.fetch32_at(wi0 + 2 * u, sh0)) : 0;
                pk[u] = L ? ((lane + L) | (1 << 8)) : kPackBig;
                if (fast && c + u < c1) {
                    ftb->len[(c + u) * kWave + lane - fo] = (uint8_t)L;
                    ftb->sym[(c + u) * kWave + lane - fo] = (uint16_t)sym;
                }
            }
            if (c == 0) CGIC_STAMP3(17);
            for (int r = 0; r < t.dbl_rounds; ++r) {
#pragma unroll
                for (int u = 0; u < kU; ++u) {
                    if (c + u < c1) {                           // wave-uniform
                        const int nx = pk[u] & 0xFF;
                        const int o = __shfl(pk[u], nx & 63, kWave);
                        if (nx < kWave) pk[u] = o + (pk[u] & 0xFF00);   // new next | (count + count on the way); counts <= 64
                    }
                }
            }
            if (c == 0) CGIC_STAMP3(18);
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (fast && c + u < c1) ftb->fn[(c + u) * kWave + lane - fo] = (uint16_t)pk[u];
                if (c + u < c1) {
                    const int o = __shfl(pk[u], F & 63, kWave);
                    if (F < kWave) { C += o >> 8; F = o & 0xFF; }
                    F = F >= kPackBig ? kBig : F - kWave;
                }
            }
            if (c == 0) CGIC_STAMP3(19);
            if (c == kU) CGIC_STAMP3(20);
        }
    }
    sh->F[wave][lane] = F;
    sh->C[wave][lane] = C;
    CGIC_STAMP3(3);
    __syncthreads();
    CGIC_STAMP3(4);
    if (bf_out) {
        // the range's function for all 64 entry offsets at once: lane = entry offset, the waves' functions applied in order
        if (wave == 0) {
            int cur = lane, cnt = 0;
            if (active) {
                for (int v = w0; v < w0 + nw; ++v) {
                    if (cur < kWave) { cnt += sh->C[v][cur]; cur = sh->F[v][cur]; }
                }
            }
            bf_out[lane] = ((uint32_t)cnt << 8) | (uint32_t)(cur < kWave ? cur : 0xFF);
        }
        return;
    }
    if (sync && sg == 0) {
        // split stream, one launch: this range's function (same walk as above) goes out, the true entry comes back
        if (wave == 0) {
            int cur = lane, cnt = 0;
            if (active) {
                for (int v = w0; v < w0 + nw; ++v) {
                    if (cur < kWave) { cnt += sh->C[v][cur]; cur = sh->F[v][cur]; }
                }
            }
            part_exchange(part, nparts, ((uint32_t)cnt << 8) | (uint32_t)(cur < kWave ? cur : 0xFF), *sync);
        }
        __syncthreads();
        e_in = sync->s_entry[0];
        n_in = sync->s_entry[1];
    }
    // ---- pass B: true entry offset + output index of this wave's range, and the segment's exit
    int e = e_in, n = n_in;
    int e_out = e_in, n_out = n_in;
    if (active) {
        // a single segment needs the walk only up to this wave (the last wave adds its own count for the total);
        // with more segments every wave walks all of them to know where the next segment starts
        const int vend = nseg > 1 ? w0 + nw : w0 + k;
        for (int v = w0; v < vend; ++v) {
            if (v == w0 + k) { e = e_out; n = n_out; }
            if (e_out >= kWave) break;
            n_out += sh->C[v][e_out];
            e_out = sh->F[v][e_out];
        }
        if (nseg == 1) { e = e_out; n = n_out; }
        e = __builtin_amdgcn_readfirstlane(e);       // wave-uniform by construction; tell the compiler
        n = __builtin_amdgcn_readfirstlane(n);
        e_out = __builtin_amdgcn_readfirstlane(e_out);
        n_out = __builtin_amdgcn_readfirstlane(n_out);
        if (nseg == 1) {
            if (k == nw - 1 && lane == 0) *count_out = e < kWave ? n + sh->C[wave][e] : n;
        } else if (sg == nseg - 1 && k == 0 && lane == 0) {
            *count_out = n_out;
        }
    }
    // ---- pass C: decode the range from its true entry offset
    CGIC_STAMP3(5);
    if (active && c1 > c0 && e < kWave && fast) {
        // lane-per-chunk: the chunk functions stored by pass A give every chunk's entry offset and output
        // index with one uniform LDS read each; then lane j walks chunk c0+j's codeword chain through
        // the cached lengths -- up to 64 chains at once on the vector unit instead of one chain at a
        // time on the CU's single scalar unit (13 us -> ~2 us for the medium stream of a 256x256 image)
        __builtin_amdgcn_wave_barrier();
        for (int cb = c0; cb < c1; cb += kWave) {
            int my_e = kBig, my_n = 0;
            const int cend = cb + kWave < c1 ? cb + kWave : c1;
            for (int c = cb; c < cend && e < kWave; ++c) {
                if (lane == c - cb) { my_e = e; my_n = n; }
                const int v = ftb->fn[c * kWave + e - fo];
                n += v >> 8;
                e = (v & 0xFF) >= kPackBig ? kBig : (v & 0xFF) - kWave;
            }
            const int c = cb + lane;
            int i = my_e, o = my_n;
            while (i < kWave) {
                const int L = ftb->len[c * kWave + i - fo];
                if (L == 0) break;
                if (o < cap) put(o, (int)ftb->sym[c * kWave + i - fo]);
                ++o;
                i += L;
            }
        }
    } else if (active && c1 > c0 && e < kWave) {
        // big streams: lookups for kU chunks issued together, then the scalar chains one after the other
        if (!bw.covers((c0 + kU + 2) * kWave) || 1 + ((c0 * kWave) >> 3) < bw.wb) bw.fill(1 + ((c0 * kWave) >> 3));
        for (int c = c0; c < c1 && e < kWave; c += kU) {
            if (!bw.covers((c + kU + 2) * kWave)) bw.fill(1 + ((c * kWave) >> 3));
            int Ls[kU], syms[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                syms[u] = 0;
                Ls[u] = c + u < c1 ? codeword_at(t, lut, bw, (c + u) * kWave + lane, nbits, &syms[u]) : 0;
            }
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                if (c + u < c1 && e < kWave) {
                    // scalar chain: everything derives from readfirstlane / readlane results, so the
                    // compiler keeps it in SGPRs (a value loaded from LDS would make the loop an
                    // exec-masked vector loop: 16 instructions + a 64-bit vector shift per symbol)
                    unsigned long long starts = 0;
                    int i = __builtin_amdgcn_readfirstlane(e);
                    while (i < kWave) {
                        const int Li = __builtin_amdgcn_readlane(Ls[u], i);
                        if (Li == 0) { i = kBig; break; }
                        starts |= 1ull << i;
                        i += Li;
                    }
                    e = i >= kBig / 2 ? kBig : i - kWave;
                    const int rank = __popcll(starts & ((1ull << lane) - 1ull));
                    if (((starts >> lane) & 1ull) && n + rank < cap) put(n + rank, syms[u]);
                    n += __popcll(starts);
                }
            }
        }
    }
    CGIC_STAMP3(6);
    __syncthreads();
    CGIC_STAMP3(7);
    e_in = e_out;
    n_in = n_out;
    if (nseg > 1 && e_in >= kWave) break;            // the stream ended inside this segment (uniform: all waves computed it)
    }   // segments
}

// single-stream decode (HuffmanCoding / BinaryCoding .decompress_string): one 1024-thread workgroup,
// the same segmented decoder; tables with codes longer than 64 bits take the one-wave serial path
__global__ __launch_bounds__(kDecThreads) void decode_stream_kernel(DecodeOneArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    __shared__ int s_count;
    uint32_t *lut = sm;
    uint32_t *win = lut + kDecLutMax;
    SegShared *seg = reinterpret_cast<SegShared *>(win + kDecWaves * kSegWinWords);
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    if (tid == 0) s_count = 0;
    load_lut(a.tab, lut);
    if (a.tab.n_nodes <= kLdsTrieNodes) {
        int32_t *ltrie = reinterpret_cast<int32_t *>(seg + 1);
        for (int i = tid; i < 2 * a.tab.n_nodes; i += kDecThreads) ltrie[i] = a.tab.child[i];
        a.tab.child = ltrie;
    }
    __syncthreads();
    const int nb = (int)a.nbytes;
    if (nb <= 0) {
        if (tid == 0) *a.count = -1;                              // empty file -> None (:158-159)
        return;
    }
    const int cap = a.cap > 0x7FFFFFFF ? 0x7FFFFFFF : (int)a.cap;
    int64_t *dst = a.syms;
    auto put = [&](int k, int sym) { dst[k] = sym; };
    if (a.tab.max_len <= 64) {
        int nw = (nb + 15) >> 4;
        nw = nw < 1 ? 1 : (nw > kDecWaves ? kDecWaves : nw);
        decode_segmented(a.tab, lut, win + wave * kSegWinWords, seg, a.in, nb, (int)a.in[0], 0, wave < nw ? nw : 0, wave, cap,
                         put, &s_count);
    } else if (wave == 0) {
        int overflow = 0;
        WaveDecoder d{a.tab, lut, win, a.in, nb, 0, 0};
        int cnt = d.run(cap, put, &overflow);
        if (__any(overflow)) cnt = cap + 1;
        if (lane == 0) s_count = cnt;
    }
    __syncthreads();
    if (tid == 0) *a.count = s_count > cap ? (int64_t)CGIC_ERR_CAPACITY : (int64_t)s_count;
}

// -------------------------------------------------------------------------------------------
// decompress = two launches                                               model.py:269-397
//   decode_streams_kernel  grid (3 streams, B), 1024 threads: every (stream, image) gets a whole
//       CU -- measured on MI355X the segmented decoder is bound by ONE CU's LDS-crossbar
//       (ds_bpermute) and scalar-issue throughput, so spreading an image over three CUs beats
//       sharing 16 waves between its streams.  Symbols go to a u16 workspace.
//   merge_kernel           grid (4 row bands, B), 256 threads: mask streams -> bitsets + popcount
//       prefixes (recomputed per band, they are tiny), then scatter + x1/x2/x4 merge + gather for
//       the band's rows.
// -------------------------------------------------------------------------------------------
struct DecodeArgs {
    TableDev tab;
    const uint8_t *in;
    int64_t slot;
    const int32_t *nbytes;
    int64_t h, w;
    int stream_mask;
    uint16_t *dsym;          // [B, n_c + n_m + n_f]
    int32_t *dcount;         // [B, 3]: >=0 count, -1 empty file (None), -2 not sent, -3 overflow
    int32_t *status;         // [B] zeroed here for the merge kernel's atomicMin
    int parts;               // workgroups per stream in the split-stream launches (kDecParts), else 1
    uint32_t *bf;            // [B, 3, parts, 64] range functions written by decode_functions_kernel
    unsigned int *tick;      // [B, 3] ticket slots of decode_split_kernel (NULL: the two-launch form)
};

__global__ __launch_bounds__(kDecThreads) void decode_streams_kernel(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    __shared__ int s_count;
    uint32_t *lut = sm;                                 // [kDecLutMax]
    uint32_t *win = lut + kDecLutMax;                   // [16][kSegWinWords] (slow mode: 1 x kWinWords)
    SegShared *seg = reinterpret_cast<SegShared *>(win + kDecWaves * kSegWinWords);
    const int tid = threadIdx.x, lane = lane_id(), wave = tid >> 6;
    // grid (B, 3), image-fastest, the usually longest stream first (medium, fine, coarse): see compress_streams_kernel
    const int s = blockIdx.y == 0 ? 1 : blockIdx.y == 1 ? 2 : 0;
    const int64_t b = blockIdx.x;
    const int64_t n_c = (a.h >> 2) * (a.w >> 2), n_m = (a.h >> 1) * (a.w >> 1), n_f = a.h * a.w;
    const int64_t off = s == 0 ? 0 : (s == 1 ? n_c : n_c + n_m);
    const int cap = (int)(s == 0 ? n_c : (s == 1 ? n_m : n_f));
    if (s == 0 && tid == 0 && a.status) a.status[b] = 0;
    int32_t *dc = a.dcount + b * 3 + s;
    const uint8_t *in = a.in + (b * CGIC_NUM_STREAMS + s) * a.slot;
    // one wave of independent loads: stream length, header byte (slot memory is always readable),
    // and the decode LUT
    __shared__ int s_nb, s_pad;
    if (tid == 0) {
        s_nb = ((a.stream_mask >> s) & 1) ? a.nbytes[b * CGIC_NUM_STREAMS + s] : -2;
        s_pad = in[0];
        s_count = 0;
    }
    CGIC_STAMP3(0);
    load_lut(a.tab, lut);
    // the decode trie (codes longer than the LUT window) next to the LUT: a speculative bit offset that
    // lands on a long-code prefix must not cost a global-memory round trip per trie step
    if (a.tab.n_nodes <= kLdsTrieNodes) {
        int32_t *ltrie = reinterpret_cast<int32_t *>(seg + 1);
        for (int i = tid; i < 2 * a.tab.n_nodes; i += kDecThreads) ltrie[i] = a.tab.child[i];
        a.tab.child = ltrie;
    }
    __syncthreads();
    CGIC_STAMP3(1);
    const int nb = s_nb;
    if (nb <= 0) {
        if (tid == 0) *dc = nb == 0 ? -1 : -2;
        return;
    }
    uint16_t *dst = a.dsym + b * (n_c + n_m + n_f) + off;
    auto put = [&](int k, int sym) { dst[k] = (uint16_t)sym; };
    if (a.tab.max_len <= 64) {
        int nw = (nb + 15) >> 4;                         // no wave below ~2 chunks
        nw = nw < 1 ? 1 : (nw > kDecWaves ? kDecWaves : nw);
        FastTables *ft = reinterpret_cast<FastTables *>(reinterpret_cast<int32_t *>(seg + 1) + 2 * kLdsTrieNodes);
        // two instantiations: the single-segment one (every stream of a 256x256 image at the usual ratios) keeps its
        // tighter code -- the segment loop cost it 0.8 us
        if (nb <= kFastChunks * 8)
            decode_segmented<false>(a.tab, lut, win + wave * kSegWinWords, seg, in, nb, s_pad, 0, wave < nw ? nw : 0, wave, cap, put,
                                    &s_count, ft);
        else
            decode_segmented<true>(a.tab, lut, win + wave * kSegWinWords, seg, in, nb, s_pad, 0, wave < nw ? nw : 0, wave, cap, put,
                                   &s_count, ft);
    } else if (wave == 0) {
        // tables with codes longer than 64 bits: one wave, serial chain
        int overflow = 0;
        WaveDecoder d{a.tab, lut, win, in, nb, 0, 0};
        int cnt = d.run(cap, put, &overflow);
        if (__any(overflow)) cnt = cap + 1;
        if (lane == 0) s_count = cnt;
    }
    __syncthreads();
    CGIC_STAMP3(8);
    if (tid == 0) *dc = s_count > cap ? -3 : s_count;
}

// ---- split streams (large grids: the streams of a 768x768 tile are ~1350 chunks; one workgroup needs nine 160-chunk
// segments one after the other -- 108 us for 8 tiles on an otherwise idle GPU).  Two launches over the grid
// (3 * parts, B), no workgroup waits for another:
//   decode_functions_kernel: every workgroup builds the FUNCTION of its chunk range -- for each of the 64 bit offsets
//     at which its first codeword might start: where the first codeword of the NEXT range starts and how many
//     symbols lie between -- by the same pointer-doubling pass, and stores the 64 entries;
//   decode_parts_kernel: composes the functions of the ranges before its own and decodes its range from the true offset.
__device__ __forceinline__ void decode_part_prologue(DecodeArgs &a, uint32_t *lut, SegShared *seg, int s, int64_t b,
                                                     const uint8_t *in, int *s_nb, int *s_pad)
{
    const int tid = threadIdx.x;
    if (tid == 0) {
        *s_nb = ((a.stream_mask >> s) & 1) ? a.nbytes[b * CGIC_NUM_STREAMS + s] : -2;
        *s_pad = in[0];
    }
    load_lut(a.tab, lut);
    if (a.tab.n_nodes <= kLdsTrieNodes) {
        int32_t *ltrie = reinterpret_cast<int32_t *>(seg + 1);
        for (int i = tid; i < 2 * a.tab.n_nodes; i += kDecThreads) ltrie[i] = a.tab.child[i];
        a.tab.child = ltrie;
    }
}

__global__ __launch_bounds__(kDecThreads) void decode_functions_kernel(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    __shared__ int s_count, s_nb, s_pad;
    uint32_t *lut = sm;
    uint32_t *win = lut + kDecLutMax;
    SegShared *seg = reinterpret_cast<SegShared *>(win + kDecWaves * kSegWinWords);
    const int tid = threadIdx.x, wave = tid >> 6;
    const int s = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
    const int64_t b = blockIdx.y;
    if (part == a.parts - 1) return;                             // nobody reads the last range's function
    const uint8_t *in = a.in + (b * CGIC_NUM_STREAMS + s) * a.slot;
    decode_part_prologue(a, lut, seg, s, b, in, &s_nb, &s_pad);
    __syncthreads();
    const int nb = s_nb;
    uint32_t *bf = a.bf + ((b * 3 + s) * a.parts + part) * kWave;
    if (nb <= 0) {
        if (tid < kWave) bf[tid] = 0xFFu;                        // no stream: every entry is "past the end"
        return;
    }
    decode_segmented<false>(a.tab, lut, win + wave * kSegWinWords, seg, in, nb, s_pad, 0, kDecWaves, wave, 0,
                            [](int, int) {}, &s_count, (FastTables *)nullptr, part, a.parts, 0, 0, bf);
}

__global__ __launch_bounds__(kDecThreads) void decode_parts_kernel(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    __shared__ int s_count, s_nb, s_pad;
    __shared__ uint32_t s_bf[kDecParts * kWave];
    __shared__ int s_entry[2];
    uint32_t *lut = sm;
    uint32_t *win = lut + kDecLutMax;
    SegShared *seg = reinterpret_cast<SegShared *>(win + kDecWaves * kSegWinWords);
    const int tid = threadIdx.x, wave = tid >> 6;
    const int s = blockIdx.x / a.parts, part = blockIdx.x % a.parts;
    const int64_t b = blockIdx.y;
    const int64_t n_c = (a.h >> 2) * (a.w >> 2), n_m = (a.h >> 1) * (a.w >> 1), n_f = a.h * a.w;
    const int64_t off = s == 0 ? 0 : (s == 1 ? n_c : n_c + n_m);
    const int cap = (int)(s == 0 ? n_c : (s == 1 ? n_m : n_f));
    if (s == 0 && part == 0 && tid == 0 && a.status) a.status[b] = 0;
    int32_t *dc = a.dcount + b * 3 + s;
    const uint8_t *in = a.in + (b * CGIC_NUM_STREAMS + s) * a.slot;
    if (tid == 0) s_count = 0;
    const uint32_t *bf = a.bf + ((b * 3 + s) * a.parts) * kWave;
    for (int i = tid; i < part * kWave; i += kDecThreads) s_bf[i] = bf[i];
    decode_part_prologue(a, lut, seg, s, b, in, &s_nb, &s_pad);
    __syncthreads();
    const int nb = s_nb;
    if (nb <= 0) {
        if (tid == 0 && part == a.parts - 1) *dc = nb == 0 ? -1 : -2;
        return;
    }
    if (tid == 0) {
        int e = 0, n = 0;
        for (int g = 0; g < part && e < kWave; ++g) {
            const uint32_t v = s_bf[g * kWave + e];
            n += (int)(v >> 8);
            e = (v & 0xFF) == 0xFF ? kBig : (int)(v & 0xFF);
        }
        s_entry[0] = e; s_entry[1] = n;
    }
    __syncthreads();
    uint16_t *dst = a.dsym + b * (n_c + n_m + n_f) + off;
    auto put = [&](int k, int sym) { dst[k] = (uint16_t)sym; };
    FastTables *ft = reinterpret_cast<FastTables *>(reinterpret_cast<int32_t *>(seg + 1) + 2 * kLdsTrieNodes);
    decode_segmented<true>(a.tab, lut, win + wave * kSegWinWords, seg, in, nb, s_pad, 0, kDecWaves, wave, cap, put, &s_count, ft,
                           part, a.parts, s_entry[0], s_entry[1]);
    __syncthreads();
    if (tid == 0 && part == a.parts - 1) *dc = s_count > cap ? -3 : s_count;      // the last part knows the total
}

// Both of the above in ONE launch: pass A once, the range functions exchanged between the workgroups of a stream through
// global memory and a ticket slot (part_exchange) instead of a kernel boundary -- the launch gap, the second staging of the
// LUT / trie / bit windows and the second pass A go away (8 tiles of 768x768: 14.3 + 19.7 us as two launches).  Ranges
// longer than the per-position tables (tiles beyond ~800x800) still build their function with a pass of their own.
//
// The workgroups of an image (gridDim.x of them: 4 for grids up to 64x64, 24 beyond) are dealt to its streams BY STREAM
// LENGTH, on the device (the byte counts live there): every stream that was sent gets one, the rest go one at a time to the
// stream with the most bytes per workgroup.  At the usual (0.1, 0.8, 0.1) ratios the medium stream of a 256x256 image is twice
// the fine one and gets the fourth workgroup (its pass A is bound by one CU's LDS crossbar: two CUs halve it); with only the
// fine grid sent (ratio (0,0,1): 590 chunks) all four decode that one stream.  Order inside an image: medium parts, fine
// parts, coarse parts -- a part's predecessors have smaller workgroup ids.
// (scalars only, no indexed arrays, no loop in the common case: everything here stays on the scalar unit)
__device__ __forceinline__ void decode_roles(int n0, int n1, int n2, int wgs, int *q0, int *q1, int *q2)
{
    // a part per kDecPartBytes of stream (160 chunks: one pass-A round of the 16 waves), at most kDecPartsMax
    auto want = [](int n) -> int {
        if (n <= 0) return 0;
        const int p = (int)(((unsigned)n + (unsigned)kDecPartBytes - 1u) / (unsigned)kDecPartBytes);
        return p > kDecPartsMax ? kDecPartsMax : p;
    };
    int p0 = want(n0), p1 = want(n1), p2 = want(n2);
    // more than the image has workgroups (long streams on a small grid): take from the stream with the most parts
    while (p0 + p1 + p2 > wgs) {
        if (p1 >= p2 && p1 >= p0) --p1;
        else if (p2 >= p0) --p2;
        else --p0;
    }
    *q0 = p0; *q1 = p1; *q2 = p2;
}

__global__ __launch_bounds__(kDecThreads) CGIC_VGPR_CAP_DECODE void decode_split_kernel(DecodeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    __shared__ int s_count;
    __shared__ uint32_t s_fn[kWave];
    __shared__ int s_entry[2];
    uint32_t *lut = sm;
    uint32_t *win = lut + kDecLutMax;
    SegShared *seg = reinterpret_cast<SegShared *>(win + kDecWaves * kSegWinWords);
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t b = blockIdx.y;
    // who am I: every workgroup of the image derives the same split from the same three byte counts.  The counts and the
    // three header bytes are requested first, the LUT / trie staging (the same for every role) runs while they arrive.
    const uint8_t *in0 = a.in + (b * CGIC_NUM_STREAMS) * a.slot;
    const int n0 = (a.stream_mask & 1) ? a.nbytes[b * CGIC_NUM_STREAMS] : -2, n1 = (a.stream_mask & 2) ? a.nbytes[b * CGIC_NUM_STREAMS + 1] : -2,
              n2 = (a.stream_mask & 4) ? a.nbytes[b * CGIC_NUM_STREAMS + 2] : -2;
    const int pad0 = in0[0], pad1 = in0[a.slot], pad2 = in0[2 * a.slot];       // (slot memory is always readable)
    if (tid == 0) s_count = 0;
    load_lut(a.tab, lut);
    if (a.tab.n_nodes <= kLdsTrieNodes) {
        int32_t *ltrie = reinterpret_cast<int32_t *>(seg + 1);
        for (int i = tid; i < 2 * a.tab.n_nodes; i += kDecThreads) ltrie[i] = a.tab.child[i];
        a.tab.child = ltrie;
    }
    int p0, p1, p2;
    decode_roles(n0, n1, n2, (int)gridDim.x, &p0, &p1, &p2);
    int s, part = (int)blockIdx.x, nparts;
    if (part < p1) { s = 1; nparts = p1; }
    else if ((part -= p1) < p2) { s = 2; nparts = p2; }
    else if ((part -= p2) < p0) { s = 0; nparts = p0; }
    else { s = -1; nparts = 0; }
    if (blockIdx.x == 0 && tid == 0) {
        if (a.status) a.status[b] = 0;
        if (n0 <= 0) a.dcount[b * 3] = n0 == 0 ? -1 : -2;          // empty file (None) / not sent
        if (n1 <= 0) a.dcount[b * 3 + 1] = n1 == 0 ? -1 : -2;
        if (n2 <= 0) a.dcount[b * 3 + 2] = n2 == 0 ? -1 : -2;
    }
    if (s < 0) return;                                            // more workgroups than the streams are worth
    const int64_t n_c = (a.h >> 2) * (a.w >> 2), n_m = (a.h >> 1) * (a.w >> 1), n_f = a.h * a.w;
    const int64_t off = s == 0 ? 0 : (s == 1 ? n_c : n_c + n_m);
    const int cap = (int)(s == 0 ? n_c : (s == 1 ? n_m : n_f));
    int32_t *dc = a.dcount + b * 3 + s;
    const uint8_t *in = in0 + s * a.slot;
    const int nb = s == 0 ? n0 : s == 1 ? n1 : n2;                // > 0: the stream has a workgroup
    const int s_pad = s == 0 ? pad0 : s == 1 ? pad1 : pad2;
    __syncthreads();
    uint16_t *dst = a.dsym + b * (n_c + n_m + n_f) + off;
    auto put = [&](int k, int sym) { dst[k] = (uint16_t)sym; };
    FastTables *ft = reinterpret_cast<FastTables *>(reinterpret_cast<int32_t *>(seg + 1) + 2 * kLdsTrieNodes);
    if (nparts == 1) {
        // the stream is this workgroup's alone (like decode_streams_kernel)
        int nw = (nb + 15) >> 4;                         // no wave below ~2 chunks
        nw = nw < 1 ? 1 : (nw > kDecWaves ? kDecWaves : nw);
        if (nb <= kFastChunks * 8)
            decode_segmented<false>(a.tab, lut, win + wave * kSegWinWords, seg, in, nb, s_pad, 0, wave < nw ? nw : 0, wave, cap, put,
                                    &s_count, ft);
        else
            decode_segmented<true>(a.tab, lut, win + wave * kSegWinWords, seg, in, nb, s_pad, 0, wave < nw ? nw : 0, wave, cap, put,
                                   &s_count, ft);
        __syncthreads();
        if (tid == 0) *dc = s_count > cap ? -3 : s_count;
        return;
    }
    PartSync ps{a.bf + (b * 3 + s) * kDecPartsMax * kWave, a.tick + (b * 3 + s) * kTicketStride, s_entry};
    int nbits = s_pad == 0 ? 0 : (nb - 1) * 8 - s_pad;
    nbits = nbits < 0 ? 0 : nbits;
    const int nchunks = (nbits + kWave - 1) / kWave, per_part = (nchunks + nparts - 1) / nparts;
    if (per_part <= kFastChunks) {
        decode_segmented<true>(a.tab, lut, win + wave * kSegWinWords, seg, in, nb, s_pad, 0, kDecWaves, wave, cap, put, &s_count, ft,
                               part, nparts, 0, 0, nullptr, &ps);
    } else {
        decode_segmented<false>(a.tab, lut, win + wave * kSegWinWords, seg, in, nb, s_pad, 0, kDecWaves, wave, 0,
                                [](int, int) {}, &s_count, (FastTables *)nullptr, part, nparts, 0, 0, s_fn);
        __syncthreads();
        if (wave == 0) part_exchange(part, nparts, s_fn[tid], ps);
        if (tid == 0) s_count = 0;
        __syncthreads();
        decode_segmented<true>(a.tab, lut, win + wave * kSegWinWords, seg, in, nb, s_pad, 0, kDecWaves, wave, cap, put, &s_count, ft,
                               part, nparts, s_entry[0], s_entry[1]);
    }
    __syncthreads();
    if (tid == 0 && part == nparts - 1) *dc = s_count > cap ? -3 : s_count;      // the last part knows the total
}

// -------------------------------------------------------------------------------------------
// Self-synchronising decoder: ONE workgroup per image decodes all three index streams (round 2, second half).
//
// The kernels above find the codeword boundaries of a chunk for EVERY possible entry offset (64 speculative starts per
// 64-bit chunk, pointer doubling, function composition): worst-case parallel time, but ~11 000 wave instructions, 16 waves
// and 132 KB of LDS per 1 KB stream -- one workgroup per CU and stream for ~11 us.  With several batches in flight
// (pipeline.LaneStream) what counts is the CU time a launch consumes, not its latency, and that was the second largest
// item of the step (decode + merge 14.7 of 46.8 us).
//
// Here a lane owns a 64-bit chunk and simply GUESSES its entry offset (0), decodes the chunk serially from the LUT and
// notes where it ran out (exit offset = entry of the next chunk) and how many symbols it saw.  Then every lane compares
// its guess with its predecessor's exit and decodes again if they differ, until a sweep changes nothing.  The first chunk
// of a stream is right from the start, so chunk k is right after at most k sweeps (exact for any input: the fixpoint is
// unique); a Huffman stream re-synchronises within a few codewords, so in practice the second sweep already changes
// nothing for almost every chunk.  A block scan of the counts gives the output positions and a last walk stores the
// symbols.  Work: ~4 serial walks of ~9 LUT lookups per chunk instead of 64 starts x 6 doubling rounds; footprint: a
// 256-thread workgroup and LUT + stream bytes of LDS (41 KB for a 256x256 image), three or more workgroups per CU.
// Worst case (a stream built never to re-synchronise) degrades to one sweep per chunk -- still exact.
// Entry offsets stay below 64 because a codeword is at most 64 bits (max_len <= 64; longer tables take the serial path).
// -------------------------------------------------------------------------------------------
constexpr int kSsEnd = 0xFF;            // "the stream ended before this chunk" as an entry / exit offset
#ifndef CGIC_SS_THREADS_SMALL
#define CGIC_SS_THREADS_SMALL 256
#endif

struct SsLayout {                       // wave-uniform description of the image's three streams
    int nbits[3];                       // payload bits
    int c[4];                           // first chunk of stream s; c[3] = total
    int off[3];                         // byte offset of the stream's copy in the LDS stage (16-byte aligned)
};
// (selects, not indexed loads: a private array indexed by a per-lane value would live in scratch memory)
__device__ __forceinline__ int sel3(int s, int v0, int v1, int v2) { return s == 0 ? v0 : (s == 1 ? v1 : v2); }

// 32 payload bits starting at bit `pos` (0..63) of the 96-bit big-endian window d0:d1:d2
__device__ __forceinline__ uint32_t ss_bits(uint32_t d0, uint32_t d1, uint32_t d2, int pos)
{
    const uint32_t hi = pos < 32 ? d0 : d1, lo = pos < 32 ? d1 : d2;
    const uint32_t r = __builtin_amdgcn_alignbit(hi, lo, 32u - ((uint32_t)pos & 31u));      // shift amount is mod 32
    return (pos & 31) ? r : hi;
}

// One serial walk over chunk g from `entry`: returns the exit offset (entry of the next chunk; kSsEnd when the stream
// ends in this chunk), *count = codewords that start in the chunk.  A lone wave issues an instruction every 5-9 cycles,
// so the walk is priced by its instruction count per codeword: 32-bit funnel shifts instead of 64-bit vector shifts, and
// the end-of-stream checks only in the stream's last two chunks (`rem` < 128).
template <bool WRITE, typename Put>
__device__ __forceinline__ int ss_walk(const TableDev &t, const uint32_t *lut, const uint8_t *stage, const SsLayout &L, int g,
                                       int entry, int *count, Put put)
{
    const int s = (g >= L.c[1]) + (g >= L.c[2]);
    const int ch = g - sel3(s, L.c[0], L.c[1], L.c[2]);
    const int rem = sel3(s, L.nbits[0], L.nbits[1], L.nbits[2]) - 64 * ch;            // payload bits from the start of this chunk
    *count = 0;
    if (entry == kSsEnd) return kSsEnd;
    // payload byte 8*ch sits at stage byte 1 + 8*ch (byte 0 is the pad count): five aligned words, shifted by one byte
    const uint32_t *q = reinterpret_cast<const uint32_t *>(stage + sel3(s, L.off[0], L.off[1], L.off[2]) + 8 * ch);
    const uint32_t r0 = __builtin_bswap32(q[0]), r1 = __builtin_bswap32(q[1]), r2 = __builtin_bswap32(q[2]),
                   r3 = __builtin_bswap32(q[3]), r4 = __builtin_bswap32(q[4]);
    const uint32_t d0 = __builtin_amdgcn_alignbit(r0, r1, 24), d1 = __builtin_amdgcn_alignbit(r1, r2, 24),
                   d2 = __builtin_amdgcn_alignbit(r2, r3, 24), d3 = __builtin_amdgcn_alignbit(r3, r4, 24);      // payload bits 0..127
    const int LB = t.lut_bits;
    int pos = entry, n = 0;
    if (t.max_len <= LB) {
        // The common case: every code fits the LUT window.  `w` holds the payload from `pos` on, left-aligned, and is
        // shifted by each code length.  Shifting loses bits at the bottom, so the window is rebuilt once, when the walk
        // crosses bit 32 (valid bits left >= 64 - 44).  The loop is PREDICATED, not divergent: every lane runs the same
        // ~14 instructions until no lane of the wave has a codeword left (one wave-uniform branch per trip); as a
        // structured loop with early exits it cost ~750 cycles per codeword -- a lone wave pays ~20 cycles for every
        // exec-mask update and taken branch, not for the arithmetic.
        const unsigned long long W01 = ((unsigned long long)d0 << 32) | d1, W23 = ((unsigned long long)d2 << 32) | d3;
        unsigned long long w = pos ? (W01 << pos) | (W23 >> (64 - pos)) : W01;
        const int sh = 64 - LB;
        bool ended = false;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int lim = rem < (half ? 64 : 32) ? rem : (half ? 64 : 32);
            while (__builtin_amdgcn_ballot_w64(pos < lim) != 0) {
                const bool go = pos < lim;
                const uint32_t e = lut[(uint32_t)(w >> sh)];
                int len = (int)(e & 0xFF);
                const bool ok = go && len != 0 && pos + len <= rem;      // a code of this table that ends inside the stream
                ended = ended || (go && !ok);                              // else: trailing partial codeword / not a code
                if (WRITE) { if (ok) put(s, n, (int)(e >> 8)); }
                len = ok ? len : 0;
                w <<= len;
                pos = (go && !ok) ? (1 << 20) : pos + len;
                n += ok ? 1 : 0;
            }
            if (half == 0) {
                const unsigned long long W12 = ((unsigned long long)d1 << 32) | d2, W3 = (unsigned long long)d3 << 32;
                const int p = (pos - 32) & 63;
                w = p ? (W12 << p) | (W3 >> (64 - p)) : W12;
            }
        }
        *count = n;
        return ended || pos < 64 ? kSsEnd : pos - 64;             // pos < 64: the stream ended at a codeword boundary in here
    }
    while (pos < 64) {
        if (pos >= rem) { *count = n; return kSsEnd; }
        const uint32_t bits = ss_bits(d0, d1, d2, pos);
        const uint32_t e = lut[bits >> (32 - LB)];
        int len = (int)(e & 0xFF), sym = (int)(e >> 8);
        if (len == 0) {
            if (sym == 0xFFFFFF) { *count = n; return kSsEnd; }
            const unsigned long long w0 = ((unsigned long long)d0 << 32) | d1, w1 = ((unsigned long long)d2 << 32) | d3;
            const unsigned long long win = pos ? (w0 << pos) | (w1 >> (64 - pos)) : w0;
            int node = sym, k = LB;
            sym = -1;
            while (pos + k < rem && k < 64) {
                const int c = t.child[2 * node + (int)((win >> (63 - k)) & 1ull)];
                ++k;
                if (c == INT32_MIN) break;
                if (c < 0) { sym = ~c; break; }
                node = c;
            }
            if (sym < 0) { *count = n; return kSsEnd; }
            len = k;
        }
        if (pos + len > rem) { *count = n; return kSsEnd; }          // trailing partial codeword: dropped by the reference
        if (WRITE) put(s, n, sym);
        pos += len;
        ++n;
    }
    *count = n;
    return pos - 64;
}

__global__ __launch_bounds__(kDecThreads) void decode_image_kernel(DecodeArgs a, int stage_cap, int chunk_cap)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    __shared__ int s_scan[kDecWaves + 1];
    __shared__ int s_base[4];
    uint32_t *lut = sm;                                             // [1 << lut_bits]
    uint8_t *stage = reinterpret_cast<uint8_t *>(lut + (1 << a.tab.lut_bits));
    uint8_t *ent = stage + stage_cap, *ext = ent + chunk_cap, *cnt = ext + chunk_cap;
    const int tid = threadIdx.x, T = blockDim.x, lane = lane_id(), wave = tid >> 6, nw = T >> 6;
    const int64_t b = blockIdx.x;
    CGIC_STAMP3(0);
    const uint8_t *in0 = a.in + (b * CGIC_NUM_STREAMS) * a.slot;
    int nb[3];
#pragma unroll
    for (int s = 0; s < 3; ++s) nb[s] = (a.stream_mask >> s & 1) ? a.nbytes[b * CGIC_NUM_STREAMS + s] : -2;
    const int pad[3] = {in0[0], in0[a.slot], in0[2 * a.slot]};      // (slot memory is always readable)
    {   // LUT -> LDS: eight 16-byte loads in flight per thread (a plain strided loop pays one memory round trip per trip:
        // 32 trips of a 256-thread workgroup = 25 us)
        const uint4 *gl = reinterpret_cast<const uint4 *>(a.tab.lut);
        uint4 *dl = reinterpret_cast<uint4 *>(lut);
        const int n4 = (1 << a.tab.lut_bits) >> 2;
        if (n4 == 0 && tid < (1 << a.tab.lut_bits)) lut[tid] = a.tab.lut[tid];      // (a table of two symbols: a 2-entry LUT)
        for (int base = 0; base < n4; base += 8 * T) {
            uint4 v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = base + k * T + tid;
                v[k] = i < n4 ? gl[i] : uint4{0u, 0u, 0u, 0u};
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int i = base + k * T + tid;
                if (i < n4) dl[i] = v[k];
            }
        }
    }
    const int64_t n_c = (a.h >> 2) * (a.w >> 2), n_m = (a.h >> 1) * (a.w >> 1), n_f = a.h * a.w;
    const int cap[3] = {(int)n_c, (int)n_m, (int)n_f};
    SsLayout L;
    int stage_used = 0;
    bool fits = true;
    L.c[0] = 0;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (nb[s] > a.slot) nb[s] = (int)a.slot;                    // (a corrupt length cannot reach beyond the slot)
        int bits = nb[s] <= 0 || pad[s] == 0 ? 0 : (nb[s] - 1) * 8 - pad[s];        // remove_padding :131-138; text[:-0] is empty
        bits = bits < 0 ? 0 : bits;
        L.nbits[s] = bits;
        L.c[s + 1] = L.c[s] + ((bits + 63) >> 6);
        L.off[s] = stage_used;
        stage_used += bits ? (((bits + 7) >> 3) + 1 + 24 + 15) & ~15 : 0;          // header + payload + the two words a walk reads past its chunk
    }
    const int C = L.c[3];
    if (stage_used > stage_cap || C > chunk_cap) fits = false;     // more bits than the grids can hold symbols: overflow
    if (tid == 0) {
        if (a.status) a.status[b] = 0;
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (nb[s] <= 0) a.dcount[b * 3 + s] = nb[s] == 0 ? -1 : -2;             // empty file (None) / not sent
            else if (!fits) a.dcount[b * 3 + s] = -3;
    }
    if (!fits) return;
    CGIC_STAMP3(1);
    // stage the stream bytes (16-byte copies; the tail beyond the stream is never interpreted: every walk checks `rem`)
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (!L.nbits[s]) continue;
        const int words = ((((L.nbits[s] + 7) >> 3) + 1 + 24 + 15) & ~15) >> 4;
        const uint4 *g = reinterpret_cast<const uint4 *>(in0 + s * a.slot);
        uint4 *d = reinterpret_cast<uint4 *>(stage + L.off[s]);
        const int lim = (int)(a.slot >> 4);                         // stay inside the slot
        for (int i = tid; i < words; i += T) d[i] = i < lim ? g[i] : uint4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    CGIC_STAMP3(2);
    // blocked ownership: lane tid owns chunks [g0, g1)
    const int R = (C + T - 1) / T;
    const int g0 = tid * R < C ? tid * R : C, g1 = g0 + R < C ? g0 + R : C;
    auto is_first = [&](int g) { return g == L.c[0] || g == L.c[1] || g == L.c[2]; };
    auto nop = [](int, int, int) {};
    {
        int prev = 0;
        for (int g = g0; g < g1; ++g) {
            // the guess: the smallest offset in the residue class the code lengths allow (0 when their gcd is 1); inside the
            // lane's own run the predecessor is known
            int e = prev;
            if (is_first(g)) e = 0;
            else if (g == g0) {
                const int sg = (g >= L.c[1]) + (g >= L.c[2]);
                const int chg = g - sel3(sg, L.c[0], L.c[1], L.c[2]);
                const int m = (64 * chg) % a.tab.len_gcd;
                e = m ? a.tab.len_gcd - m : 0;
            }
            int n;
            prev = ss_walk<false>(a.tab, lut, stage, L, g, e, &n, nop);
            ent[g] = (uint8_t)e; ext[g] = (uint8_t)prev; cnt[g] = (uint8_t)n;
        }
    }
    __syncthreads();
    CGIC_STAMP3(3);
    [[maybe_unused]] int dbg_sweeps = 0;
#ifdef CGIC_PHASE_CLOCKS
    if (blockIdx.x == 0 && tid == 0) { g_phase_clk[23] = C; g_phase_clk[24] = R; int mx = 0; for (int g = 0; g < C; ++g) mx = cnt[g] > mx ? cnt[g] : mx; g_phase_clk[25] = mx; }
#endif
    for (;;) {
        ++dbg_sweeps;
        int changed = 0;
        for (int g = g0; g < g1; ++g) {
            const int e = is_first(g) ? 0 : ext[g - 1];
            if (e != ent[g]) {
                int n;
                const int x = ss_walk<false>(a.tab, lut, stage, L, g, e, &n, nop);
                ent[g] = (uint8_t)e; ext[g] = (uint8_t)x; cnt[g] = (uint8_t)n;
                changed = 1;
            }
        }
        if (!__syncthreads_or(changed)) break;
    }
    CGIC_STAMP3(4);
#ifdef CGIC_PHASE_CLOCKS
    if (blockIdx.x == 0 && tid == 0) g_phase_clk[9] = dbg_sweeps;
#endif
    // output positions: exclusive prefix of the counts over the chunks, restarted at every stream
    int mine = 0;
    for (int g = g0; g < g1; ++g) mine += cnt[g];
    int inc = wave_inclusive_scan(mine);
    if (lane == kWave - 1) s_scan[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        const int v = lane < nw ? s_scan[lane] : 0;
        const int vi = wave_inclusive_scan(v);
        if (lane < nw) s_scan[lane] = vi - v;
        if (lane == nw - 1) s_scan[kDecWaves] = vi;
    }
    __syncthreads();
    int run = s_scan[wave] + inc - mine;
    const int total = s_scan[kDecWaves];
    {
        int r = run;
        for (int g = g0; g < g1; ++g) {
            if (g == L.c[1]) s_base[1] = r;
            if (g == L.c[2]) s_base[2] = r;
            r += cnt[g];
        }
        if (tid == 0) {
            s_base[0] = 0; s_base[3] = total;
            if (L.c[2] == C) s_base[2] = total;
            if (L.c[1] == C) s_base[1] = total;
        }
    }
    __syncthreads();
    CGIC_STAMP3(5);
    uint16_t *dst = a.dsym + b * (n_c + n_m + n_f);
    for (int g = g0; g < g1; ++g) {
        const int s = (g >= L.c[1]) + (g >= L.c[2]);
        const int at = run - s_base[s];
        const int cap_s = sel3(s, cap[0], cap[1], cap[2]);
        uint16_t *dst_s = dst + sel3(s, 0, (int)n_c, (int)(n_c + n_m)) + at;
        int n;
        ss_walk<true>(a.tab, lut, stage, L, g, (int)ent[g], &n,
                      [&](int, int k, int sym) { if (at + k < cap_s) dst_s[k] = (uint16_t)sym; });
        run += n;
    }
    CGIC_STAMP3(6);
    if (tid < 3 && sel3(tid, nb[0], nb[1], nb[2]) > 0) {
        const int n = s_base[tid + 1] - s_base[tid];
        a.dcount[b * 3 + tid] = n > sel3(tid, cap[0], cap[1], cap[2]) ? -3 : n;
    }
}

constexpr int kMergeThreads = 512;
constexpr int kMergeBands = 4;          // row bands per image at least; more for few large images (gridDim.x)

struct MergeArgs {
    const uint8_t *in;
    int64_t slot;
    const int32_t *nbytes;
    int64_t h, w;
    int mode;
    const uint16_t *dsym;
    const int32_t *dcount;
    int64_t *ind_out;
    int32_t *mc_out, *mm_out, *mf_out;
    const float *codebook;
    int K;
    float *zq;
    const float *codebook2;    // second table gathered with the same indices (post_quant_conv(codebook)), or NULL
    float *zq2;
    int32_t *status;
    int stage_sym, stage_cb;   // keep the image's decoded symbols / the codebook in LDS
    int64_t band_syms;         // u16 entries reserved for a band's own symbol ranges when stage_sym == 0
};

// NT threads per workgroup: 512 with >= 4 row bands per image (the shortest launch for one batch on an idle GPU), or 1024 with ONE
// band per image (throughput mode, grids up to 64x64): every band repeats the staging, the bitsets and the prefixes, so four bands
// of 512 threads execute 1.10 M VALU instructions per batch of 64 images where one band of 1024 executes less than half --
// instructions that, with several batches in flight, come out of the same VALU budget as the VQ's.
template <int NT>
__global__ __launch_bounds__(NT) CGIC_VGPR_CAP_MERGE void merge_kernel(MergeArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    __shared__ uint32_t scan_smem[NT / kWave + 1];
    __shared__ int s_status;
    __shared__ int s_hdr[8];            // nbytes[3], nbytes[4], dcount[0..2]
    const int tid = threadIdx.x;
    const int band = blockIdx.x;
    const int64_t b = blockIdx.y;
    const int64_t h = a.h, w = a.w, h2 = h >> 1, w2 = w >> 1, h4 = h >> 2, w4 = w >> 2;
    const int64_t n_c = h4 * w4, n_m = h2 * w2, n_f = h * w;
    const int64_t wc = (n_c + 31) >> 5, wm = (n_m + 31) >> 5;
    const int64_t nsym = n_c + n_m + n_f;
    // LDS: [codebook rows (16-B aligned)][symbols u16][raw mask-stream words][bitsets][prefixes]
    float4 *cbk = reinterpret_cast<float4 *>(sm);                                   // [K] if a.stage_cb
    uint16_t *lsym = reinterpret_cast<uint16_t *>(cbk + (a.stage_cb ? a.K : 0));   // [nsym] if a.stage_sym
    uint32_t *rawc = reinterpret_cast<uint32_t *>(lsym) + (a.stage_sym ? (nsym + 1) / 2 : 0);   // [wc + 2]
    uint32_t *rawm = rawc + wc + 2;     // [wm + 2] stream bytes incl. header, as loaded
    uint32_t *mcb = rawm + wm + 2;      // [wc] coarse mask bits, LSB first
    uint32_t *mmb = mcb + wc;           // [wm]
    uint32_t *pcb = mmb + wm;           // [wc] exclusive popcount prefix
    uint32_t *pmb = pcb + wc;           // [wm]
    const int mode = a.mode;
    // this block's rows: bands of whole coarse rows (multiples of 4 fine rows)
    const int64_t nbands = gridDim.x;
    const int64_t rows_per = ((h4 + nbands - 1) / nbands) * 4;
    const int64_t r0 = band * rows_per, r1 = r0 + rows_per < h ? r0 + rows_per : h;
    if (r0 >= h) return;
    CGIC_STAMP(10);
    const bool send_mc = mode == 0 || mode == 2 || mode == 3;
    const bool send_mm = mode == 0 || mode == 1;
    const uint8_t *in_mc = a.in + (b * CGIC_NUM_STREAMS + 3) * a.slot;
    const uint8_t *in_mm = a.in + (b * CGIC_NUM_STREAMS + 4) * a.slot;

    // ---- ONE wave of independent global loads: headers, both mask streams (slots are 16-byte
    // aligned and at least wc*4+8 / wm*4+8 bytes long), all decoded symbols, the codebook
    if (tid == 0) s_status = 0;
    if (tid < 2) s_hdr[tid] = a.nbytes[b * CGIC_NUM_STREAMS + 3 + tid];
    else if (tid < 5) s_hdr[tid] = a.dcount[b * 3 + (tid - 2)];
    if (send_mc) for (int64_t i = tid; i < wc + 2; i += NT) rawc[i] = reinterpret_cast<const uint32_t *>(in_mc)[i];
    if (send_mm) for (int64_t i = tid; i < wm + 2; i += NT) rawm[i] = reinterpret_cast<const uint32_t *>(in_mm)[i];
    const uint16_t *gsym = a.dsym + b * nsym;
    if (a.stage_sym) {
        // nsym = 21 * n_c is even; the per-image base is 4-byte aligned when nsym is even
        const uint32_t *g32 = reinterpret_cast<const uint32_t *>(gsym);
        uint32_t *l32 = reinterpret_cast<uint32_t *>(lsym);
        for (int64_t i = tid; i < (nsym + 1) / 2; i += NT) l32[i] = g32[i];
    }
    if (a.stage_cb && a.zq)
        for (int i = tid; i < a.K; i += NT) cbk[i] = reinterpret_cast<const float4 *>(a.codebook)[i];
    __syncthreads();
    CGIC_STAMP(11);

    // a mask stream must be exactly 1 + n/8 + 1 bytes with pad = 8 - n%8 (mask_coding.py:19-26)
    if (tid == 0) {
        if (send_mc && (s_hdr[0] != 2 + (n_c >> 3) || (int)(rawc[0] & 0xFF) != 8 - (int)(n_c & 7))) s_status = CGIC_ERR_INVALID;
        if (send_mm && (s_hdr[1] != 2 + (n_m >> 3) || (int)(rawm[0] & 0xFF) != 8 - (int)(n_m & 7))) s_status = CGIC_ERR_INVALID;
    }
    __syncthreads();
    if (s_status) {
        if (tid == 0 && a.status) atomicMin(&a.status[b], s_status);
        return;
    }
    // MSB-first stream bytes (after the header byte) -> LSB-first bit words
    auto stream_word = [](const uint32_t *raw, int64_t wi, int64_t nbits) -> uint32_t {
        // payload bytes 4wi..4wi+3 are stream bytes 1+4wi.. : straddle raw[wi], raw[wi+1]
        const uint64_t two = (uint64_t)raw[wi] | ((uint64_t)raw[wi + 1] << 32);
        const uint32_t pay = (uint32_t)(two >> 8);                  // 4 payload bytes, little-endian order
        // reverse the bits inside each byte: brev reverses all 32, bswap puts the bytes back
        uint32_t v = __builtin_bswap32(__brev(pay));
        const int64_t rem = nbits - wi * 32;
        if (rem < 32) v &= rem <= 0 ? 0u : ((1u << rem) - 1u);
        return v;
    };
    uint32_t cnt_c, cnt_m;
    __shared__ uint32_t s_cnt[2];
    const bool derived_mm = mode == 3 || mode == 5;       // medium mask built from the coarse one / all ones
    if (wc <= kWave && wm <= kWave && !derived_mm) {
        // small masks (<= 2048 positions): wave 0 builds the coarse bitset + prefix, wave 1 the medium one, each with
        // one wave scan -- one barrier instead of two loops + two block scans (8 barriers)
        const int lane = lane_id(), wave = tid >> 6;
        if (wave < 2) {
            const bool co = wave == 0;
            const int64_t nw_ = co ? wc : wm, nb_ = co ? n_c : n_m;
            uint32_t v = 0;
            if (lane < nw_) {
                if (co ? send_mc : send_mm) v = stream_word(co ? rawc : rawm, lane, nb_);
                else if (co && mode == 4) {                                         // ones (:355)
                    v = 0xFFFFFFFFu;
                    const int64_t rem = nb_ - (int64_t)lane * 32;
                    if (rem < 32) v &= (1u << rem) - 1u;
                }
            }
            const uint32_t c = (uint32_t)__popc(v);
            const uint32_t inc = wave_inclusive_scan(c);
            if (lane < nw_) {
                (co ? mcb : mmb)[lane] = v;
                (co ? pcb : pmb)[lane] = inc - c;
            }
            if (lane == kWave - 1) s_cnt[wave] = inc;
        }
        __syncthreads();
        cnt_c = s_cnt[0];
        cnt_m = s_cnt[1];
    } else {
        for (int64_t i = tid; i < wc; i += NT) {
            uint32_t v = 0;
            if (send_mc) v = stream_word(rawc, i, n_c);
            else if (mode == 4) {                                                   // ones (:355)
                v = 0xFFFFFFFFu;
                const int64_t rem = n_c - i * 32;
                if (rem < 32) v &= (1u << rem) - 1u;
            }
            mcb[i] = v;
        }
        __syncthreads();
        for (int64_t i = tid; i < wm; i += NT) {
            uint32_t v = 0;
            if (send_mm) v = stream_word(rawm, i, n_m);
            else if (mode == 3 || mode == 5) {
                for (int k = 0; k < 32; ++k) {
                    const int64_t j = i * 32 + k;
                    if (j >= n_m) break;
                    bool bit = true;                                                // mode 5: ones (:368)
                    if (mode == 3) {                                                // 1 - up2(mask_coarse) (:332)
                        const int64_t y = j / w2, x = j - y * w2, c = (y >> 1) * w4 + (x >> 1);
                        bit = !((mcb[c >> 5] >> (c & 31)) & 1u);
                    }
                    v |= (uint32_t)bit << k;
                }
            }
            mmb[i] = v;
        }
        __syncthreads();
        if (wc <= NT && wm <= NT) {
            // both prefixes from ONE block scan of (coarse count << 32 | medium count): three barriers instead of six
            __shared__ unsigned long long scan64[NT / kWave + 1];
            const unsigned long long c = ((unsigned long long)(tid < wc ? __popc(mcb[tid]) : 0) << 32) | (unsigned long long)(tid < wm ? __popc(mmb[tid]) : 0);
            unsigned long long tot;
            const unsigned long long ex = block_exclusive_scan(c, scan64, &tot);
            if (tid < wc) pcb[tid] = (uint32_t)(ex >> 32);
            if (tid < wm) pmb[tid] = (uint32_t)ex;
            cnt_c = (uint32_t)(tot >> 32);
            cnt_m = (uint32_t)tot;
            __syncthreads();
        } else {
        uint32_t carry = 0, total;
        for (int64_t base = 0; base < wc; base += NT) {
            const int64_t i = base + tid;
            const uint32_t c = i < wc ? (uint32_t)__popc(mcb[i]) : 0u;
            const uint32_t ex = block_exclusive_scan(c, scan_smem, &total);
            if (i < wc) pcb[i] = carry + ex;
            carry += total;
        }
        cnt_c = carry;
        carry = 0;
        for (int64_t base = 0; base < wm; base += NT) {
            const int64_t i = base + tid;
            const uint32_t c = i < wm ? (uint32_t)__popc(mmb[i]) : 0u;
            const uint32_t ex = block_exclusive_scan(c, scan_smem, &total);
            if (i < wm) pmb[i] = carry + ex;
            carry += total;
        }
        cnt_m = carry;
        __syncthreads();
        }
    }
    CGIC_STAMP(12);

    auto fine_flag = [&](int64_t y, int64_t x, bool *pbc, bool *pbm) -> bool {
        const int64_t j2 = (y >> 1) * w2 + (x >> 1), j4 = (y >> 2) * w4 + (x >> 2);
        const bool bc = (mcb[j4 >> 5] >> (j4 & 31)) & 1u;
        const bool bm = (mmb[j2 >> 5] >> (j2 & 31)) & 1u;
        *pbc = bc; *pbm = bm;
        switch (mode) {
        case 0: return (1 - (int)bm - (int)bc) == 1;                            // :280
        case 1: return !bm;                                                     // :302
        case 2: return !bc;                                                     // :320
        case 6: return true;                                                    // :380
        default: return false;
        }
    };
    // fine symbols consumed by the rows above this band (exact for any mask bits)
    uint32_t fbase = 0;
    {
        // the flag is constant over a 2x2 medium cell: walk the medium cells of the rows above (rows by wave, columns by
        // lane, no divisions) and count 4 per cell.  (The per-position loop cost the LAST band of a 768x768 tile 70 trips of
        // ~40 instructions -- the merge launch took twice as long as its first band.)
        uint32_t mine = 0;
        const int wvm = tid >> 6, lnm = tid & 63;
        for (int y2 = wvm; y2 < (int)(r0 >> 1); y2 += NT / 64)
            for (int x2 = lnm; x2 < (int)w2; x2 += 64) {
                bool bc, bm;
                mine += fine_flag(2 * y2, 2 * x2, &bc, &bm) ? 4u : 0u;
            }
        (void)block_exclusive_scan(mine, scan_smem, &fbase);
    }
    CGIC_STAMP(13);

    const uint16_t *ds_c = a.stage_sym ? lsym : gsym, *ds_m = ds_c + n_c, *ds_f = ds_m + n_m;
    const int64_t dc_c = s_hdr[2], dc_m = s_hdr[3], dc_f = s_hdr[4];
    // Large images (the symbols of the whole image do not fit LDS): this band only consumes three CONTIGUOUS rank ranges --
    // the coarse / medium ones of the mask bits inside its rows and the fine ones from fbase on.  One coalesced load of
    // those (a.band_syms entries reserved behind the prefixes) replaces two or three dependent global loads per position.
    uint16_t *bsym = reinterpret_cast<uint16_t *>(pmb + wm);
    int64_t off_c = 0, off_m = 0, off_f = 0;           // global rank of the first staged entry of each range
    if (!a.stage_sym && a.band_syms > 0) {
        auto rank_of = [&](const uint32_t *bits, const uint32_t *pre, int64_t j, int64_t nbits, uint32_t total) -> int64_t {
            if (j >= nbits) return (int64_t)total;
            return (int64_t)pre[j >> 5] + __popc(bits[j >> 5] & ((1u << (j & 31)) - 1u));
        };
        const int64_t c0 = rank_of(mcb, pcb, (r0 >> 2) * w4, n_c, cnt_c), c1 = rank_of(mcb, pcb, (r1 >> 2) * w4, n_c, cnt_c);
        const int64_t m0 = rank_of(mmb, pmb, (r0 >> 1) * w2, n_m, cnt_m), m1 = rank_of(mmb, pmb, (r1 >> 1) * w2, n_m, cnt_m);
        const int64_t f0 = fbase, f1 = f0 + (r1 - r0) * w;                    // at most every position of the band
        const int64_t nc = c1 - c0, nm = m1 - m0, nf = (f1 < n_f ? f1 : n_f) - f0;
        if (nc >= 0 && nm >= 0 && nf >= 0 && nc + nm + nf + 3 <= a.band_syms) {
            for (int64_t i = tid; i < nc; i += NT) bsym[i] = ds_c[c0 + i];
            for (int64_t i = tid; i < nm; i += NT) bsym[nc + i] = ds_m[m0 + i];
            for (int64_t i = tid; i < nf; i += NT) bsym[nc + nm + i] = ds_f[f0 + i];
            __syncthreads();
            off_c = c0; off_m = m0 - nc; off_f = f0 - nc - nm;     // ds_x[rank] == bsym[rank - off_x]
            ds_c = bsym; ds_m = bsym; ds_f = bsym;
        }
    }
    const bool has_c = mode == 0 || mode == 2 || mode == 3 || mode == 4;
    const bool has_m = mode == 0 || mode == 1 || mode == 3 || mode == 5;
    const bool has_f = mode == 0 || mode == 1 || mode == 2 || mode == 6;
    // an empty index file (None) means "all zeros" for coarse/medium (:283-290); any other
    // count must equal the number of mask ones (the reference raises a shape mismatch)
    int st = 0;
    if (dc_c == -3 || dc_m == -3 || dc_f == -3) st = CGIC_ERR_INVALID;
    if (has_c && dc_c >= 0 && dc_c != cnt_c) st = CGIC_ERR_INVALID;
    if (has_m && dc_m >= 0 && dc_m != cnt_m) st = CGIC_ERR_INVALID;
    const bool use_c = has_c && dc_c >= 0, use_m = has_m && dc_m >= 0, use_f = has_f && dc_f >= 0;

    int64_t *ind_out = a.ind_out ? a.ind_out + b * n_f : nullptr;
    float *zq = a.zq ? a.zq + b * 4 * n_f : nullptr;
    float *zq2 = a.zq2 ? a.zq2 + b * 4 * n_f : nullptr;
    uint32_t fcarry = fbase;
    int bad_index = 0;
    // One thread per QUAD of four consecutive positions of a row (w % 4 == 0): they share their coarse cell and lie in two
    // medium cells, so a quad costs one coarse and two medium rank lookups instead of four of each, and every output is one
    // 16-byte store per plane (the per-position form issued 4-byte stores 8 bytes apart).
    const int w4i = (int)w4;
    for (int64_t qbase = r0 * w4; qbase < r1 * w4; qbase += NT) {
        const int64_t q = qbase + tid;
        const bool live = q < r1 * w4;
        int64_t v[4] = {0, 0, 0, 0};
        bool bfa = false, bfb = false;
        int y = 0, xq = 0;
        if (live) {
            y = (int)q / w4i; xq = (int)q - y * w4i;                           // 32-bit divide (h*w < 2^26)
            const int64_t j4 = (int64_t)(y >> 2) * w4 + xq, j2 = (int64_t)(y >> 1) * w2 + 2 * xq;
            bool bc, bma, bmb, dummy;
            bfa = fine_flag(y, 4 * xq, &bc, &bma);
            bfb = fine_flag(y, 4 * xq + 2, &dummy, &bmb);
            int64_t vc = 0, va = 0, vb = 0;
            if (bc && use_c) vc = ds_c[(int64_t)pcb[j4 >> 5] + __popc(mcb[j4 >> 5] & ((1u << (j4 & 31)) - 1u)) - off_c];
            if (use_m) {
                // j2 is even: both cells sit in the same bitset word
                const uint32_t word = mmb[j2 >> 5], below = word & ((1u << (j2 & 31)) - 1u);
                const int64_t rk = (int64_t)pmb[j2 >> 5] + __popc(below) - off_m;
                if (bma) va = ds_m[rk];
                if (bmb) vb = ds_m[rk + (bma ? 1 : 0)];
            }
            v[0] = v[1] = vc + va;
            v[2] = v[3] = vc + vb;
            const int64_t i = (int64_t)y * w + 4 * xq;
            if (a.mc_out && (y & 3) == 0) a.mc_out[b * n_c + j4] = bc;
            if (a.mm_out && (y & 1) == 0) *reinterpret_cast<int2 *>(a.mm_out + b * n_m + j2) = make_int2(bma, bmb);
            if (a.mf_out) *reinterpret_cast<int4 *>(a.mf_out + b * n_f + i) = make_int4(bfa, bfa, bfb, bfb);
        }
        uint32_t ftotal;
        uint32_t frank = block_exclusive_scan((bfa ? 2u : 0u) + (bfb ? 2u : 0u), scan_smem, &ftotal) + fcarry;
        if (live) {
            if (bfa) {
                if (use_f) {                                                    // t[t==1] = decoded (:292)
                    if ((int64_t)frank < dc_f) v[0] += ds_f[(int64_t)frank - off_f];
                    if ((int64_t)frank + 1 < dc_f) v[1] += ds_f[(int64_t)frank + 1 - off_f];
                }
                frank += 2;
            }
            if (bfb && use_f) {
                if ((int64_t)frank < dc_f) v[2] += ds_f[(int64_t)frank - off_f];
                if ((int64_t)frank + 1 < dc_f) v[3] += ds_f[(int64_t)frank + 1 - off_f];
            }
            const int64_t i = (int64_t)y * w + 4 * xq;
            if (ind_out) {                                                      // sum of the three grids (:293)
                reinterpret_cast<longlong2 *>(ind_out + i)[0] = make_longlong2(v[0], v[1]);
                reinterpret_cast<longlong2 *>(ind_out + i)[1] = make_longlong2(v[2], v[3]);
            }
            if (zq || zq2) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (v[k] < 0 || v[k] >= a.K) { bad_index = 1; v[k] = 0; }
            }
            if (zq) {
                float4 e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = a.stage_cb ? cbk[v[k]] : reinterpret_cast<const float4 *>(a.codebook)[v[k]];   // exact rows (:391-392)
                *reinterpret_cast<float4 *>(zq + i) = make_float4(e[0].x, e[1].x, e[2].x, e[3].x);
                *reinterpret_cast<float4 *>(zq + n_f + i) = make_float4(e[0].y, e[1].y, e[2].y, e[3].y);
                *reinterpret_cast<float4 *>(zq + 2 * n_f + i) = make_float4(e[0].z, e[1].z, e[2].z, e[3].z);
                *reinterpret_cast<float4 *>(zq + 3 * n_f + i) = make_float4(e[0].w, e[1].w, e[2].w, e[3].w);
            }
            if (zq2) {
                float4 e[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) e[k] = reinterpret_cast<const float4 *>(a.codebook2)[v[k]];       // 16 KB table: L1 / L2 hits
                *reinterpret_cast<float4 *>(zq2 + i) = make_float4(e[0].x, e[1].x, e[2].x, e[3].x);
                *reinterpret_cast<float4 *>(zq2 + n_f + i) = make_float4(e[0].y, e[1].y, e[2].y, e[3].y);
                *reinterpret_cast<float4 *>(zq2 + 2 * n_f + i) = make_float4(e[0].z, e[1].z, e[2].z, e[3].z);
                *reinterpret_cast<float4 *>(zq2 + 3 * n_f + i) = make_float4(e[0].w, e[1].w, e[2].w, e[3].w);
            }
        }
        fcarry += ftotal;
    }
    CGIC_STAMP(14);
    // the last band sees the total number of fine positions
    if (r1 == h && has_f && (dc_f >= 0 ? dc_f != (int64_t)fcarry : fcarry != 0)) st = CGIC_ERR_INVALID;
    if (bad_index) s_status = CGIC_ERR_INVALID;
    __syncthreads();
    if (tid == 0 && a.status && (st || s_status)) atomicMin(&a.status[b], st ? st : s_status);
}

__global__ void gather_kernel(const int64_t *__restrict__ ind, int64_t B, int64_t hw,
                              const float *__restrict__ cb, int K, float *__restrict__ out,
                              int32_t *__restrict__ status)
{
    const int64_t n = B * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t v = ind[i];
        const int64_t b = i / hw, p = i - b * hw;
        if (v < 0 || v >= K) { if (status) status[b] = CGIC_ERR_INVALID; v = 0; }
        const float4 e = reinterpret_cast<const float4 *>(cb)[v];
        float *o = out + b * 4 * hw + p;
        o[0] = e.x; o[hw] = e.y; o[2 * hw] = e.z; o[3 * hw] = e.w;
    }
}

static const int kModeStreams[7] = {0x1f, 0x16, 0x0d, 0x0b, 0x01, 0x02, 0x04};  // model.py:225-260

static size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

}  // namespace cgic

using namespace cgic;

#ifdef CGIC_PHASE_CLOCKS
extern "C" int cgic_debug_block_times(long long *out, int n)
{
    CGIC_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blk_t), sizeof(long long) * 2 * (size_t)n));
    return CGIC_OK;
}
extern "C" int cgic_debug_reset_span(void)
{
    long long init[4] = {0x7fffffffffffffffLL, 0, 0, 0};
    CGIC_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_phase_clk), init, sizeof(init), sizeof(long long) * 28));
    return CGIC_OK;
}
extern "C" int cgic_debug_phase_clocks(long long *out16)
{
    CGIC_HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_clk), sizeof(long long) * 32));
    return CGIC_OK;
}
#endif

extern "C" int cgic_mode_streams(int mode)
{
    CGIC_REQUIRE(mode >= 0 && mode <= 6, CGIC_ERR_INVALID, "mode %d outside 0..6", mode);
    return kModeStreams[mode];
}

extern "C" size_t cgic_stream_capacity(const cgic_table *t, int64_t n)
{
    if (!t || n < 0) return 0;
    const uint64_t bits = (uint64_t)cgic_table_max_len(t) * (uint64_t)n;
    return align16((size_t)(bits / 8 + 2) + 8);    // header + pad byte + word-store/fetch slack
}

extern "C" size_t cgic_stream_workspace_bytes(int64_t n)
{
    return n > kLdsPos ? align16((size_t)n * 4) + align16((size_t)n * 2) : 0;
}

extern "C" size_t cgic_compress_slot_bytes(const cgic_table *t, int64_t h, int64_t w)
{
    if (!t || h <= 0 || w <= 0) return 0;
    const size_t a = cgic_stream_capacity(t, h * w);
    const size_t m = align16((size_t)((h / 2) * (w / 2) / 8 + 2) + 8);
    return a > m ? a : m;
}

static size_t ws_stride(int64_t h, int64_t w) { return ((size_t)(h * w) + 7) & ~(size_t)7; }

extern "C" size_t cgic_compress_workspace_bytes(int64_t B, int64_t h, int64_t w)
{
    if (B <= 0 || h * w <= kLdsPos) return 0;
    return (size_t)B * 3 * ws_stride(h, w) * (sizeof(uint32_t) + sizeof(uint16_t));
}

static int check_grid(int64_t B, int64_t h, int64_t w, int mode)
{
    CGIC_REQUIRE(B >= 0 && h > 0 && w > 0 && h % 4 == 0 && w % 4 == 0, CGIC_ERR_INVALID,
                 "latent grid %lldx%lld must be positive multiples of 4", (long long)h, (long long)w);
    CGIC_REQUIRE(mode >= 0 && mode <= 6, CGIC_ERR_INVALID, "mode %d outside 0..6", mode);
    CGIC_REQUIRE(B <= 65535, CGIC_ERR_UNSUPPORTED, "batch %lld exceeds the grid limit", (long long)B);
    CGIC_REQUIRE(h * w < ((int64_t)1 << 26), CGIC_ERR_UNSUPPORTED, "latent grid too large");
    return CGIC_OK;
}

extern "C" int cgic_compress_streams(const cgic_table *t, const int64_t *ind, const int32_t *mask_c,
                                     const int32_t *mask_m, const int32_t *mask_f, int64_t B, int64_t h,
                                     int64_t w, int mode, uint8_t *out, int64_t slot, int32_t *nbytes,
                                     int64_t *hist, void *workspace, cgic_stream_t stream)
{
    int rc = check_grid(B, h, w, mode);
    if (rc) return rc;
    CGIC_REQUIRE(t && ind && mask_c && mask_m && mask_f && out && nbytes, CGIC_ERR_INVALID, "compress_streams: NULL argument");
    CGIC_REQUIRE(slot % 16 == 0 && (size_t)slot >= cgic_compress_slot_bytes(t, h, w), CGIC_ERR_CAPACITY,
                 "compress_streams: slot=%lld, need a multiple of 16 >= %zu", (long long)slot, cgic_compress_slot_bytes(t, h, w));
    CGIC_REQUIRE(cgic_table_num_symbols(t) <= 65536, CGIC_ERR_UNSUPPORTED, "table too large");
    CGIC_REQUIRE((uint64_t)cgic_table_max_len(t) * (uint64_t)(h * w) < 0xFFFFFF00ull, CGIC_ERR_UNSUPPORTED,
                 "compress_streams: a stream could exceed 2^32 bits");
    CGIC_REQUIRE(workspace || cgic_compress_workspace_bytes(B, h, w) == 0, CGIC_ERR_INVALID,
                 "compress_streams: workspace required for %lldx%lld grids", (long long)h, (long long)w);
    if (B == 0) return CGIC_OK;
    CompressArgs a;
    rc = table_device_view(t, &a.tab);
    if (rc) return rc;
    a.ind = ind; a.mc = mask_c; a.mm = mask_m; a.mf = mask_f; a.h = h; a.w = w;
    a.stream_mask = kModeStreams[mode];
    a.out = out; a.slot = slot; a.nbytes = nbytes;
    CGIC_REQUIRE(!hist || cgic_table_num_symbols(t) <= kLdsPos, CGIC_ERR_UNSUPPORTED, "compress_streams: hist needs n <= %d", kLdsPos);
    a.hist = (unsigned long long *)hist;
    a.ws_stride = (int64_t)ws_stride(h, w);
    a.ws_end = (uint32_t *)workspace;
    a.ws_sym = workspace ? (uint16_t *)((char *)workspace + (size_t)B * 3 * ws_stride(h, w) * sizeof(uint32_t)) : nullptr;
    // Streams beyond kLdsPos positions are split over workgroups of at most ~kLdsPos positions each (a 768x768 tile: fine
    // 36 864 positions -> 5 parts, medium 9216 -> 2) when the launch is small enough for the ticket pool; every part stages
    // 2 bytes per position of its range in dynamic LDS (static: 57.5 KB).
    a.parts[0] = a.parts[1] = a.parts[2] = 1;
    a.tick = nullptr;
    int64_t longest = 0;                  // positions the longest workgroup stages
    const bool split = B * 6 <= (int64_t)(16384 / 4);
    for (int g = 0; g < 3; ++g) {
        const int64_t npos = (h >> (2 - g)) * (w >> (2 - g));
#ifndef CGIC_ENC_PART_POS
#define CGIC_ENC_PART_POS 4096       // positions per part: measured 8192 -> 14.1 us, 4096 -> 12.2 us, 3072 -> 12.2 us (8 tiles of 768x768)
#endif
        int64_t P = split && npos > kLdsPos ? (npos + CGIC_ENC_PART_POS - 1) / CGIC_ENC_PART_POS : 1;
        P = P > kEncMaxParts ? kEncMaxParts : P;
#ifdef CGIC_ENC_PARTS_MAX
        P = P > CGIC_ENC_PARTS_MAX ? CGIC_ENC_PARTS_MAX : P;
#endif
        a.parts[g] = (int)P;
        const int64_t per = P > 1 ? (((npos + P - 1) / P) + 3) & ~(int64_t)3 : npos;
        if (npos > kLdsPos && per > longest) longest = per;
    }
    size_t dyn = 0;
    a.stage_positions = 0;
    if (longest > 0 && (size_t)longest * 2 <= 96 * 1024) {
        dyn = (((size_t)longest + 3) / 4 * 4 * 2 + 64 + 15) & ~(size_t)15;       // whole 4-entry groups (+ slack)
        a.stage_positions = longest;
        { int rc_ = ensure_dynamic_lds((const void *)compress_streams_kernel, dyn); if (rc_) return rc_; }
    } else if (longest > 0) {
        a.parts[0] = a.parts[1] = a.parts[2] = 1;                 // no staging room: the round-by-round form, unsplit
    }
    if (a.parts[0] + a.parts[1] + a.parts[2] > 3) {
        rc = acquire_tickets((hipStream_t)stream, (int)(B * 6), &a.tick);
        if (rc) return rc;
    }
    const unsigned jobs = (unsigned)(a.parts[0] + a.parts[1] + a.parts[2]) + 2u + (hist ? 1u : 0u);
    hipLaunchKernelGGL(compress_streams_kernel, a.tick ? dim3(jobs, (unsigned)B) : dim3((unsigned)B, jobs), dim3(kEncThreads), dyn,
                       (hipStream_t)stream, a);
    return launch_check("compress_streams_kernel");
}

extern "C" int cgic_encode_stream(const cgic_table *t, const void *syms, int elem_bytes, int64_t n, uint8_t *out,
                                  int64_t cap, int32_t *nbytes, void *workspace, cgic_stream_t stream)
{
    CGIC_REQUIRE(t && out && nbytes && (syms || n == 0), CGIC_ERR_INVALID, "encode_stream: NULL argument");
    CGIC_REQUIRE(elem_bytes == 8 || elem_bytes == 4, CGIC_ERR_INVALID, "encode_stream: elem_bytes must be 4 or 8");
    CGIC_REQUIRE(n >= 0 && n < ((int64_t)1 << 26), CGIC_ERR_UNSUPPORTED, "encode_stream: n out of range");
    CGIC_REQUIRE(((uintptr_t)out & 3) == 0, CGIC_ERR_INVALID, "encode_stream: out must be 4-byte aligned");
    CGIC_REQUIRE(workspace || n <= kLdsPos, CGIC_ERR_INVALID, "encode_stream: workspace required for n > %d", kLdsPos);
    CGIC_REQUIRE((uint64_t)cgic_table_max_len(t) * (uint64_t)n < 0xFFFFFF00ull, CGIC_ERR_UNSUPPORTED,
                 "encode_stream: the stream could exceed 2^32 bits");
    EncodeOneArgs a;
    int rc = table_device_view(t, &a.tab);
    if (rc) return rc;
    a.syms = syms; a.elem_bytes = elem_bytes; a.n = n; a.out = out; a.cap = cap; a.nbytes = nbytes;
    a.ws_end = (uint32_t *)workspace;
    a.ws_sym = workspace ? (uint16_t *)((char *)workspace + align16((size_t)n * 4)) : nullptr;
    hipLaunchKernelGGL(encode_stream_kernel, dim3(1), dim3(kEncThreads), 0, (hipStream_t)stream, a);
    return launch_check("encode_stream_kernel");
}

extern "C" int cgic_decode_stream(const cgic_table *t, const uint8_t *in, int64_t nbytes, int64_t *syms,
                                  int64_t cap, int64_t *count, cgic_stream_t stream)
{
    CGIC_REQUIRE(t && count && (in || nbytes == 0) && (syms || cap == 0), CGIC_ERR_INVALID, "decode_stream: NULL argument");
    CGIC_REQUIRE(nbytes >= 0 && cap >= 0 && nbytes < ((int64_t)1 << 28), CGIC_ERR_INVALID, "decode_stream: size out of range");
    DecodeOneArgs a;
    int rc = table_device_view(t, &a.tab);
    if (rc) return rc;
    a.in = in; a.nbytes = nbytes; a.syms = syms; a.cap = cap; a.count = count;
    size_t lds = sizeof(uint32_t) * (kDecLutMax + kDecWaves * kSegWinWords) + sizeof(SegShared) + sizeof(int32_t) * 2 * kLdsTrieNodes;
    if (lds < sizeof(uint32_t) * (kDecLutMax + kWinWords)) lds = sizeof(uint32_t) * (kDecLutMax + kWinWords);
    { int rc_ = ensure_dynamic_lds((const void *)decode_stream_kernel, (size_t)lds); if (rc_) return rc_; }
    hipLaunchKernelGGL(decode_stream_kernel, dim3(1), dim3(kDecThreads), lds, (hipStream_t)stream, a);
    return launch_check("decode_stream_kernel");
}

static const size_t kLdsBudget = 150 * 1024;

static std::atomic<int> g_decode_mode{CGIC_DECODE_AUTO};
extern "C" int cgic_set_decode_mode(int mode)
{
    CGIC_REQUIRE(mode == CGIC_DECODE_AUTO || mode == CGIC_DECODE_LATENCY || mode == CGIC_DECODE_THROUGHPUT, CGIC_ERR_INVALID,
                 "set_decode_mode: mode %d", mode);
    return g_decode_mode.exchange(mode);
}

extern "C" size_t cgic_decompress_workspace_bytes(int64_t B, int64_t h, int64_t w)
{
    if (B <= 0 || h <= 0 || w <= 0) return 0;
    const size_t per = (size_t)((h / 4) * (w / 4) + (h / 2) * (w / 2) + h * w);
    return align16((size_t)B * per * sizeof(uint16_t)) + align16((size_t)B * 3 * sizeof(int32_t))
           + align16((size_t)B * 3 * kDecPartsMax * kWave * sizeof(uint32_t));                  // split-stream functions
}

extern "C" int cgic_decompress_streams(const cgic_table *t, const uint8_t *in, int64_t slot, const int32_t *nbytes,
                                       int64_t B, int64_t h, int64_t w, int mode, int64_t *ind_out,
                                       int32_t *mask_c_out, int32_t *mask_m_out, int32_t *mask_f_out,
                                       const float *codebook, int K, int e_dim, float *z_q, const float *codebook2,
                                       float *z_q2, int32_t *status,
                                       void *workspace, int decoder, cgic_stream_t stream)
{
    int rc = check_grid(B, h, w, mode);
    if (rc) return rc;
    CGIC_REQUIRE(t && in && nbytes && workspace, CGIC_ERR_INVALID, "decompress_streams: NULL argument");
    CGIC_REQUIRE(decoder == CGIC_DECODE_AUTO || decoder == CGIC_DECODE_LATENCY || decoder == CGIC_DECODE_THROUGHPUT, CGIC_ERR_INVALID,
                 "decompress_streams: decoder %d", decoder);
    // which prefix decoder: a property of THIS call (AUTO = the process default of cgic_set_decode_mode)
    const int dec_mode = decoder != CGIC_DECODE_AUTO ? decoder : g_decode_mode.load();
    CGIC_REQUIRE(slot % 16 == 0 && slot >= 16 && slot < ((int64_t)1 << 28), CGIC_ERR_INVALID,
                 "decompress_streams: slot must be a multiple of 16 below 2^28");
    CGIC_REQUIRE(!z_q || (codebook && e_dim == 4 && K > 0), CGIC_ERR_UNSUPPORTED,
                 "decompress_streams: fused gather needs a [K,4] codebook");
    CGIC_REQUIRE(!z_q2 || (codebook2 && e_dim == 4 && K > 0), CGIC_ERR_UNSUPPORTED,
                 "decompress_streams: the second gather needs a [K,4] table");
    CGIC_REQUIRE(cgic_table_num_symbols(t) <= 65536, CGIC_ERR_UNSUPPORTED, "table too large");
    // the merge writes four positions per store
    CGIC_REQUIRE(((reinterpret_cast<uintptr_t>(ind_out) | reinterpret_cast<uintptr_t>(z_q) | reinterpret_cast<uintptr_t>(z_q2) |
                   reinterpret_cast<uintptr_t>(mask_f_out)) & 15) == 0 && (reinterpret_cast<uintptr_t>(mask_m_out) & 7) == 0,
                 CGIC_ERR_INVALID, "decompress_streams: outputs must be 16-byte aligned");
    if (B == 0) return CGIC_OK;
    const size_t per = (size_t)((h / 4) * (w / 4) + (h / 2) * (w / 2) + h * w);
    hipStream_t s = (hipStream_t)stream;
    DecodeArgs d;
    rc = table_device_view(t, &d.tab);
    if (rc) return rc;
    d.in = in; d.slot = slot; d.nbytes = nbytes; d.h = h; d.w = w; d.stream_mask = kModeStreams[mode];
    d.dsym = (uint16_t *)workspace;
    d.dcount = (int32_t *)((char *)workspace + align16((size_t)B * per * sizeof(uint16_t)));
    d.status = status;
    // Streams are split over workgroups that exchange range functions (decode_split_kernel): 4 workgroups per image for grids
    // up to 64x64 (a 256x256 image), 24 beyond, dealt to the streams by length on the device.  Tables with codes longer than
    // 64 bits take the one-wave path of decode_streams_kernel; batches beyond the ticket ring keep the older forms.
    const bool large = h * w > 64 * 64;
    d.parts = large && d.tab.max_len <= 64 ? kDecParts : 1;
    d.bf = (uint32_t *)((char *)d.dcount + align16((size_t)B * 3 * sizeof(int32_t)));
    size_t lds_d = sizeof(uint32_t) * (kDecLutMax + kDecWaves * kSegWinWords) + sizeof(SegShared) + sizeof(int32_t) * 2 * kLdsTrieNodes
                   + sizeof(FastTables);
    if (lds_d < sizeof(uint32_t) * (kDecLutMax + kWinWords)) lds_d = sizeof(uint32_t) * (kDecLutMax + kWinWords);
    if (lds_d > 48 * 1024)
        { int rc_ = ensure_dynamic_lds((const void *)decode_streams_kernel, (size_t)lds_d); if (rc_) return rc_; }
    d.tick = nullptr;
    // The self-synchronising one-workgroup-per-image decoder when the worst case of the grid fits its LDS: bits <= symbols
    // the three grids can hold x the longest code.  (Longer inputs are an overflow on any path.)
    bool ss = false;
#ifndef CGIC_DEC_NO_SS
    if (d.tab.max_len <= 64 && dec_mode == CGIC_DECODE_THROUGHPUT) {
        const size_t bits_cap = per * (size_t)d.tab.max_len + 3 * 64;
        const size_t stage_cap = align16(bits_cap / 8 + 3 * 48), chunk_cap = align16(bits_cap / 64 + 8);
        const size_t lds_ss = sizeof(uint32_t) * ((size_t)1 << d.tab.lut_bits) + stage_cap + 3 * chunk_cap;
        if (lds_ss <= kLdsBudget) {
            ss = true;
            { int rc_ = ensure_dynamic_lds((const void *)decode_image_kernel, lds_ss); if (rc_) return rc_; }
            const int T = large ? kDecThreads : CGIC_SS_THREADS_SMALL;
            hipLaunchKernelGGL(decode_image_kernel, dim3((unsigned)B), dim3(T), lds_ss, s, d, (int)stage_cap, (int)chunk_cap);
            rc = launch_check("decode_image_kernel");
        }
    }
#endif
    if (ss) {
    } else
#ifdef CGIC_DEC_TWO_LAUNCH
    if (false) {
#else
    if (d.tab.max_len <= 64 && B * 3 <= (int64_t)(16384 / 4)) {
#endif
        rc = acquire_tickets(s, (int)(B * 3), &d.tick);
        if (rc) return rc;
        if (lds_d > 48 * 1024)
            { int rc_ = ensure_dynamic_lds((const void *)decode_split_kernel, (size_t)lds_d); if (rc_) return rc_; }
#ifndef CGIC_DEC_WGS_SMALL
#define CGIC_DEC_WGS_SMALL 4
#endif
#ifndef CGIC_DEC_WGS_LARGE
#define CGIC_DEC_WGS_LARGE 24
#endif
        hipLaunchKernelGGL(decode_split_kernel, dim3(large ? CGIC_DEC_WGS_LARGE : CGIC_DEC_WGS_SMALL, (unsigned)B), dim3(kDecThreads), lds_d, s, d);
        rc = launch_check("decode_split_kernel");
    } else if (d.parts > 1) {
        if (lds_d > 48 * 1024) {
            { int rc_ = ensure_dynamic_lds((const void *)decode_functions_kernel, (size_t)lds_d); if (rc_) return rc_; }
            { int rc_ = ensure_dynamic_lds((const void *)decode_parts_kernel, (size_t)lds_d); if (rc_) return rc_; }
        }
        hipLaunchKernelGGL(decode_functions_kernel, dim3(3 * d.parts, (unsigned)B), dim3(kDecThreads), lds_d, s, d);
        rc = launch_check("decode_functions_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(decode_parts_kernel, dim3(3 * d.parts, (unsigned)B), dim3(kDecThreads), lds_d, s, d);
        rc = launch_check("decode_parts_kernel");
    } else {
        hipLaunchKernelGGL(decode_streams_kernel, dim3((unsigned)B, 3), dim3(kDecThreads), lds_d, s, d);
        rc = launch_check("decode_streams_kernel");
    }
    if (rc) return rc;
    MergeArgs m;
    m.in = in; m.slot = slot; m.nbytes = nbytes; m.h = h; m.w = w; m.mode = mode;
    m.dsym = d.dsym; m.dcount = d.dcount; m.ind_out = ind_out;
    m.mc_out = mask_c_out; m.mm_out = mask_m_out; m.mf_out = mask_f_out;
    m.codebook = codebook; m.K = K; m.zq = z_q; m.codebook2 = codebook2; m.zq2 = z_q2; m.status = status;
    const size_t wc = (size_t)(((h / 4) * (w / 4) + 31) / 32), wm = (size_t)(((h / 2) * (w / 2) + 31) / 32);
    size_t lds_m = (3 * (wc + wm) + 4) * sizeof(uint32_t);
    CGIC_REQUIRE(lds_m <= kLdsBudget, CGIC_ERR_UNSUPPORTED, "decompress_streams: grid too large for the mask bitsets");
    m.stage_cb = (z_q && lds_m + (size_t)K * 16 <= 64 * 1024) ? 1 : 0;
    if (m.stage_cb) lds_m += (size_t)K * 16;
    m.stage_sym = (per % 2 == 0 && lds_m + per * 2 + 4 <= 64 * 1024) ? 1 : 0;
    if (m.stage_sym) lds_m += ((per + 1) / 2) * 4;
    // mask-stream slots must cover the word-wise staging reads
    CGIC_REQUIRE((size_t)slot >= (wm + 2) * 4, CGIC_ERR_CAPACITY, "decompress_streams: slot smaller than a mask stream");
    // 4 bands per image fill the GPU at B = 64; a few large tiles get more (every band re-derives the mask prefixes,
    // so not more than needed): ~256 workgroups in all, at least 2 coarse rows per band
    int64_t nbands = kMergeBands;
    {
        const int64_t h4 = h >> 2;
#ifndef CGIC_MERGE_MINROWS
#define CGIC_MERGE_MINROWS 1
#endif
#ifndef CGIC_MERGE_WGS
#define CGIC_MERGE_WGS 256
#endif
        while (nbands * B < CGIC_MERGE_WGS && nbands * 2 <= h4 / CGIC_MERGE_MINROWS) nbands *= 2;
    }
    // the image's symbols do not fit LDS: every band stages its own three rank ranges (at most 21/16 symbols per position)
    m.band_syms = 0;
    if (!m.stage_sym) {
        const int64_t rows_per = (((h >> 2) + nbands - 1) / nbands) * 4;
        const int64_t need = rows_per * w * 21 / 16 + 8;
        if (lds_m + (size_t)need * 2 <= 64 * 1024) { m.band_syms = need; lds_m += (size_t)need * 2; }
    }
    if (dec_mode == CGIC_DECODE_THROUGHPUT && !large && m.stage_sym) {
        // several batches in flight: one band of 1024 threads per image (see merge_kernel)
        if (lds_m > 48 * 1024)
            { int rc_ = ensure_dynamic_lds((const void *)merge_kernel<1024>, (size_t)lds_m); if (rc_) return rc_; }
        hipLaunchKernelGGL(merge_kernel<1024>, dim3(1u, (unsigned)B), dim3(1024), lds_m, s, m);
        return launch_check("merge_kernel");
    }
    if (lds_m > 48 * 1024)
        { int rc_ = ensure_dynamic_lds((const void *)merge_kernel<kMergeThreads>, (size_t)lds_m); if (rc_) return rc_; }
    hipLaunchKernelGGL(merge_kernel<kMergeThreads>, dim3((unsigned)nbands, (unsigned)B), dim3(kMergeThreads), lds_m, s, m);
    return launch_check("merge_kernel");
}

extern "C" int cgic_embedding_gather_f32(const int64_t *ind, int64_t B, int64_t hw, const float *codebook, int K,
                                         int e_dim, float *out, int32_t *status, cgic_stream_t stream)
{
    CGIC_REQUIRE(ind && codebook && out, CGIC_ERR_INVALID, "embedding_gather: NULL argument");
    CGIC_REQUIRE(e_dim == 4 && K > 0, CGIC_ERR_UNSUPPORTED, "embedding_gather: needs a [K,4] codebook");
    const int64_t n = B * hw;
    if (n <= 0) return CGIC_OK;
    int nblk = (int)((n + 255) / 256);
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(gather_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, ind, B, hw, codebook, K, out, status);
    return launch_check("gather_kernel");
}
