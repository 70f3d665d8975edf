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
// (round 3: this file is the ENCODE side; the decoders live in cgic_decode.hip / cgic_decode_ss.hip)
#include "cgic_coder_dev.h"

namespace cgic {

#ifdef CGIC_PHASE_CLOCKS
__device__ long long g_phase_clk[32];
__device__ long long g_blk_t[2 * 4096];
#endif

#ifndef CGIC_ENC_THREADS
#define CGIC_ENC_THREADS 512       // 1024 -> 512 in round 3: with four batches in flight 36.3 -> 35.5 us per step (smaller workgroups find a CU sooner); alone +0.5 us
#endif
constexpr int kEncThreads = CGIC_ENC_THREADS;         // x kEncItems = 4096 positions per scan round: one round per 256x256 stream
constexpr int kEncItems = 4;            // consecutive positions per thread per scan round
#ifndef CGIC_LDS_POS
#define CGIC_LDS_POS 8192
#endif
constexpr int kLdsPos = CGIC_LDS_POS;           // streams up to this many positions keep phase-A results in LDS
constexpr int kLdsPosSmall = 4096;              // ... and the small instantiation of the compress kernel (grids up to 64x64)

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
    const EncStorage st = count <= (uint32_t)kLdsPos ? st_lds : st_glob;      // (only the kLdsPos instantiation of the kernel gets here)
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
    int combine;             // unsplit streams: workgroups per image = fine | medium | coarse + both masks | histogram
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

// LP: positions the static LDS arrays hold.  kLdsPos (8192: 48 KB) in general; kLdsPosSmall (4096: 24 KB, 32 KB with the code
// table) when no grid of the launch has more positions than that -- a 256x256 image -- because what such a latency-bound workgroup
// costs the kernels of OTHER batches in flight is the LDS it holds: 33.6 -> 32.7 us per step at four lanes (round 3).
template <int LP>
__device__ __forceinline__ void compress_streams_body(const CompressArgs &a, const Blk blk)
{
    __shared__ __attribute__((aligned(16))) uint32_t lds_end[LP];
    __shared__ uint16_t lds_sym[LP];
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
    const int64_t b = jobs_fastest ? blk.y : blk.x;
    CGIC_STAMP2(0);
    CGIC_SPAN_BEGIN();
#ifdef CGIC_PHASE_CLOCKS      // dev: per-workgroup (start, end) for tools/probes/probe_compress_blocks.py
    const unsigned int dbg_lin = (unsigned int)b * 16 + (jobs_fastest ? blk.x : blk.y);      // (image, job in launch order) like the probe expects
    if (threadIdx.x == 0 && dbg_lin < 4096) g_blk_t[2 * dbg_lin] = wall_clock64();
    struct DbgEnd { unsigned int lin; __device__ ~DbgEnd() { if (threadIdx.x == 0 && lin < 4096) g_blk_t[2 * lin + 1] = wall_clock64(); } } dbg_end{dbg_lin};
#endif
    const int64_t h = a.h, w = a.w;
    // one job = one stream of one image (or a part of it), or the image's usage histogram
    auto run_job = [&](const int s, const int part, const int nparts) {
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
        int rc;
        if (s < 3) {
            TableDev tab = a.tab;
            // code table -> LDS (length + code lookups then cost an LDS access, not an L2 round trip each)
            if (tab.n <= kLdsTable && tab.words == 1) {
                for (int i = threadIdx.x; i < tab.n; i += kEncThreads) { lds_len[i] = tab.len[i]; lds_code[i] = tab.code[i]; }
                tab.len = lds_len;
                tab.code = lds_code;
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
                rc = encode_huffman_stream_long(tab, mypos, sym_at, lds_stage, st, st_glob, out, a.slot, sh == 0 ? ind + pos0 : nullptr,
                                                sh == 0 ? mask + pos0 : nullptr, &ex);
            } else if (npos <= LP) rc = encode_huffman_stream(tab, npos, sym_at, st, out, a.slot);
            else if (a.stage_positions >= npos)
                rc = encode_huffman_stream_long(tab, npos, sym_at, lds_stage, st, st_glob, out, a.slot, sh == 0 ? ind : nullptr, sh == 0 ? mask : nullptr);
            else rc = encode_huffman_stream(tab, npos, sym_at, st_glob, out, a.slot);
        } else {
            const int sh = s == 3 ? 2 : 1;
            const int64_t npos = (h >> sh) * (w >> sh);
            const int32_t *mask = (s == 3 ? a.mc : a.mm) + b * npos;   // grain_mask[k].flatten() (:230-231)
            rc = encode_binary_stream(npos, [&](int64_t i) { return (int)mask[i]; }, out, a.slot);
        }
        if (threadIdx.x == 0 && part == nparts - 1) *nb = rc < 0 ? rc - 10 : rc;     // errors are CGIC_ERR_* - 10 (-1 means "not written")
    };
    int y = (int)(jobs_fastest ? blk.x : blk.y);
    if (a.combine) {
        // unsplit streams (grids up to kLdsPos positions): the fine stream, the medium stream, ONE workgroup for the three short
        // streams one after the other (coarse indices, the two masks), and the histogram if asked for.  As six workgroups per
        // image the launch spent its first ~4 us handing out 1024-thread workgroups (~100 per us), three of them for a few
        // hundred positions each.  (The histogram in the same workgroup as the short streams made that one the long pole:
        // 9.5 us alone against 7.6.)
        if (y < 2) {
            run_job(2 - y, 0, 1);
        } else if (y == 2) {
            run_job(0, 0, 1);
            __syncthreads();
            run_job(3, 0, 1);
            __syncthreads();
            run_job(4, 0, 1);
        } else {
            run_job(CGIC_NUM_STREAMS, 0, 1);
        }
    } else {
        int s, part = 0, nparts = 1;
        if (y < a.parts[2]) { s = 2; part = y; nparts = a.parts[2]; }
        else if ((y -= a.parts[2]) < a.parts[1]) { s = 1; part = y; nparts = a.parts[1]; }
        else if ((y -= a.parts[1]) < a.parts[0]) { s = 0; part = y; nparts = a.parts[0]; }
        else { y -= a.parts[0]; s = y == 0 ? 4 : y == 1 ? 3 : 5; }
        run_job(s, part, nparts);
    }
    CGIC_SPAN_END();
}

template <int LP>
__global__ __launch_bounds__(kEncThreads) CGIC_VGPR_CAP_COMPRESS void compress_streams_kernel(CompressArgs a)
{
    compress_streams_body<LP>(a, own_blk());
}

// several shape groups in one launch (cgic_common.h: launch groups); split streams keep their own ticket slots per group
__global__ __launch_bounds__(kEncThreads) CGIC_VGPR_CAP_COMPRESS void compress_streams_grouped_kernel(Grouped<CompressArgs> g)
{
    Blk blk;
    const CompressArgs &a = g.a[group_locate(g, &blk)];
    compress_streams_body<kLdsPos>(a, blk);
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
        { int rc_ = ensure_dynamic_lds((const void *)compress_streams_kernel<kLdsPos>, dyn); if (rc_) return rc_; }
    } else if (longest > 0) {
        a.parts[0] = a.parts[1] = a.parts[2] = 1;                 // no staging room: the round-by-round form, unsplit
    }
    if (a.parts[0] + a.parts[1] + a.parts[2] > 3) {
        rc = acquire_tickets((hipStream_t)stream, (int)(B * 6), &a.tick);
        if (rc) return rc;
    }
    // every index stream fits the static LDS arrays and nothing is split: the short jobs of an image share one workgroup
    a.combine = (!a.tick && h * w <= kLdsPos) ? 1 : 0;
    const unsigned jobs = a.combine ? 3u + (hist ? 1u : 0u) : (unsigned)(a.parts[0] + a.parts[1] + a.parts[2]) + 2u + (hist ? 1u : 0u);
    if (a.combine && h * w <= kLdsPosSmall && (!hist || cgic_table_num_symbols(t) <= kLdsPosSmall)) {
        const dim3 grid_s((unsigned)B, jobs);
        hipStream_t s_ = (hipStream_t)stream;
        return launch_or_record(KID_NONE, grid_s, dim3(kEncThreads), 0, a, s_, [=] {
            hipLaunchKernelGGL(compress_streams_kernel<kLdsPosSmall>, grid_s, dim3(kEncThreads), 0, s_, a);
            return launch_check("compress_streams_kernel"); });
    }
    const dim3 grid = a.tick ? dim3(jobs, (unsigned)B) : dim3((unsigned)B, jobs);
    hipStream_t s = (hipStream_t)stream;
    return launch_or_record(KID_COMPRESS, grid, dim3(kEncThreads), dyn, a, s, [=] {
        hipLaunchKernelGGL(compress_streams_kernel<kLdsPos>, grid, dim3(kEncThreads), dyn, s, a);
        return launch_check("compress_streams_kernel"); });
}

static int compress_grouped_launch(const GroupRec *const *recs, int n, hipStream_t s)
{
    Grouped<CompressArgs> g;
    size_t lds;
    int rc = fill_grouped(recs, n, &g, &lds);
    if (rc) return rc;
    if (lds) { rc = ensure_dynamic_lds((const void *)compress_streams_grouped_kernel, lds); if (rc) return rc; }
    hipLaunchKernelGGL(compress_streams_grouped_kernel, dim3(g.start[kMaxGroups]), dim3(kEncThreads), lds, s, g);
    return launch_check("compress_streams_grouped_kernel");
}
static GroupedRegistrar reg_compress(KID_COMPRESS, compress_grouped_launch);

extern "C" int cgic_encode_stream(const cgic_table *t, const void *syms, int elem_bytes, int64_t n, uint8_t *out,
                                  int64_t cap, int32_t *nbytes, void *workspace, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_encode_stream");
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
