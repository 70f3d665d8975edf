// cgic_decode.hip -- the DECODE side of the entropy coder (split off cgic_coder.hip in round 3):
//   prefix decoding of the index streams (indices_coding.py:131-168, mask_coding.py:59-96) by the latency-mode kernels
//   (decode_stream_kernel, decode_streams_kernel, decode_split_kernel), mask -> index scatter, x2/x4 merge and embedding gather
//   (model.py:269-397: merge_kernel, gather_kernel), and the host entry points cgic_decode_stream / cgic_decompress_streams /
//   cgic_embedding_gather_f32.  The throughput-mode decoder (decode_image_kernel) is in cgic_decode_ss.hip.
#include "cgic_coder_dev.h"

#include <atomic>

namespace cgic {

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
// segments one after the other -- 108 us for 8 tiles on an otherwise idle GPU).  The workgroups of a stream each build the
// FUNCTION of their chunk range -- for each of the 64 bit offsets at which its first codeword might start: where the first
// codeword of the NEXT range starts and how many symbols lie between -- by the pointer-doubling pass, exchange the functions,
// compose those of the ranges before their own and decode their range from the true offset.  (Rounds 1-2 did this as two
// launches, decode_functions_kernel + decode_parts_kernel; round 3 removed them: batches beyond one ticket request are cut
// into several launches of the one-launch kernel below.)
//
// ONE launch: pass A once, the range functions exchanged between the workgroups of a stream through
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

__device__ __forceinline__ void decode_split_body(const DecodeArgs &a_in, const Blk blk)
{
    DecodeArgs a = a_in;             // (the trie pointer is redirected to the LDS copy below)
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    __shared__ int s_count;
    __shared__ uint32_t s_fn[kWave];
    __shared__ int s_entry[2];
    uint32_t *lut = sm;
    uint32_t *win = lut + kDecLutMax;
    SegShared *seg = reinterpret_cast<SegShared *>(win + kDecWaves * kSegWinWords);
    const int tid = threadIdx.x, wave = tid >> 6;
    const int64_t b = blk.y;
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
    decode_roles(n0, n1, n2, (int)blk.nx, &p0, &p1, &p2);
    int s, part = (int)blk.x, nparts;
    if (part < p1) { s = 1; nparts = p1; }
    else if ((part -= p1) < p2) { s = 2; nparts = p2; }
    else if ((part -= p2) < p0) { s = 0; nparts = p0; }
    else { s = -1; nparts = 0; }
    if (blk.x == 0 && tid == 0) {
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

__global__ __launch_bounds__(kDecThreads) CGIC_VGPR_CAP_DECODE void decode_split_kernel(DecodeArgs a)
{
    decode_split_body(a, own_blk());
}

// several shape groups in one launch (cgic_common.h: launch groups)
__global__ __launch_bounds__(kDecThreads) CGIC_VGPR_CAP_DECODE void decode_split_grouped_kernel(Grouped<DecodeArgs> g)
{
    Blk blk;
    decode_split_body(g.a[group_locate(g, &blk)], blk);
}

constexpr int kMergeThreads = 512;
#ifndef CGIC_MERGE_ONE_BAND_THREADS
#define CGIC_MERGE_ONE_BAND_THREADS 512      // 1024 until round 3: 512 measured ~0.5-1 us per step better in flight
#endif
constexpr int kMergeOneBandThreads = CGIC_MERGE_ONE_BAND_THREADS;       // throughput mode: one band per image
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
// the wait of a merge workgroup that rides in the decoder's launch (decode_merge_kernel): `done` counts the image's decoder
// workgroups (each adds 1 after its last store + an agent-scope release); every band that has seen them all adds 1 itself and
// the last one to do so puts the word back to zero (nobody can still be polling then): the ticket is self-resetting.
// Called by all threads of the workgroup (contains a barrier); what follows may read what the decoder wrote.
__device__ __forceinline__ void wait_decoded(unsigned int *done, unsigned int need, unsigned int total)
{
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) __builtin_amdgcn_s_sleep(8);
        const unsigned int old = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1u == total) __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");     // ONE cache invalidate per workgroup: the loads below do not hit lines cached before the decoder's release
    }
    __syncthreads();
}

// WAIT: the workgroup shares its launch with the image's decoder workgroups: everything that does not depend on the decoded
// symbols (mask streams -> bitsets, popcount prefixes, the fine symbols consumed above the band, the codebook) runs while
// the decoder works; the symbol counts and the symbols themselves are fetched after wait_decoded()
// 16 / 8-byte nontemporal stores (round 6): the merge's outputs -- 7.7 MB per launch at B = 64 that nobody in this launch reads again --
// leave as streaming stores, like the VQ launch's indices and z_q, instead of sitting dirty in the L2s until the end-of-kernel write-back
typedef unsigned int cgic_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int cgic_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void nt_store16(void *p, const void *v)
{
#ifndef CGIC_MERGE_PLAIN_STORES
    __builtin_nontemporal_store(*reinterpret_cast<const cgic_u32x4 *>(v), reinterpret_cast<cgic_u32x4 *>(p));
#else
    *reinterpret_cast<cgic_u32x4 *>(p) = *reinterpret_cast<const cgic_u32x4 *>(v);
#endif
}
__device__ __forceinline__ void nt_store8(void *p, const void *v)
{
#ifndef CGIC_MERGE_PLAIN_STORES
    __builtin_nontemporal_store(*reinterpret_cast<const cgic_u32x2 *>(v), reinterpret_cast<cgic_u32x2 *>(p));
#else
    *reinterpret_cast<cgic_u32x2 *>(p) = *reinterpret_cast<const cgic_u32x2 *>(v);
#endif
}

template <int NT, bool WAIT = false>
__device__ __forceinline__ void merge_body(const MergeArgs &a, const Blk blk, unsigned int *done = nullptr, unsigned int need = 0,
                                           unsigned int total = 0)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t sm[];
    __shared__ uint32_t scan_smem[NT / kWave + 1];
    __shared__ int s_status;
    __shared__ int s_hdr[8];            // nbytes[3], nbytes[4], dcount[0..2]
    const int tid = threadIdx.x;
    const int band = blk.x;
    const int64_t b = blk.y;
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
    const int64_t nbands = blk.nx;
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
    else if (!WAIT && tid < 5) s_hdr[tid] = a.dcount[b * 3 + (tid - 2)];
    if (send_mc) for (int64_t i = tid; i < wc + 2; i += NT) rawc[i] = reinterpret_cast<const uint32_t *>(in_mc)[i];
    if (send_mm) for (int64_t i = tid; i < wm + 2; i += NT) rawm[i] = reinterpret_cast<const uint32_t *>(in_mm)[i];
    const uint16_t *gsym = a.dsym + b * nsym;
    if (!WAIT && a.stage_sym) {
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
        if (WAIT) wait_decoded(done, need, total);        // (the decoder's first workgroup initialises status[b])
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
    if (WAIT) {
        wait_decoded(done, need, total);
        if (tid >= 2 && tid < 5) s_hdr[tid] = a.dcount[b * 3 + (tid - 2)];
        if (a.stage_sym) {
            const uint32_t *g32 = reinterpret_cast<const uint32_t *>(gsym);
            uint32_t *l32 = reinterpret_cast<uint32_t *>(lsym);
            for (int64_t i = tid; i < (nsym + 1) / 2; i += NT) l32[i] = g32[i];
        }
        __syncthreads();
    }

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
            if (a.mm_out && (y & 1) == 0) { const int2 t2 = make_int2(bma, bmb); nt_store8(a.mm_out + b * n_m + j2, &t2); }
            if (a.mf_out) { const int4 t4 = make_int4(bfa, bfa, bfb, bfb); nt_store16(a.mf_out + b * n_f + i, &t4); }
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
                nt_store16(ind_out + i, &v[0]);
                nt_store16(ind_out + i + 2, &v[2]);
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
                const float4 o0 = make_float4(e[0].x, e[1].x, e[2].x, e[3].x), o1 = make_float4(e[0].y, e[1].y, e[2].y, e[3].y);
                const float4 o2 = make_float4(e[0].z, e[1].z, e[2].z, e[3].z), o3 = make_float4(e[0].w, e[1].w, e[2].w, e[3].w);
                nt_store16(zq + i, &o0);
                nt_store16(zq + n_f + i, &o1);
                nt_store16(zq + 2 * n_f + i, &o2);
                nt_store16(zq + 3 * n_f + i, &o3);
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

template <int NT>
__global__ __launch_bounds__(NT) CGIC_VGPR_CAP_MERGE void merge_kernel(MergeArgs a)
{
    merge_body<NT>(a, own_blk());
}

// several shape groups in one launch (cgic_common.h: launch groups)
__global__ __launch_bounds__(kMergeThreads) CGIC_VGPR_CAP_MERGE void merge_grouped_kernel(Grouped<MergeArgs> g)
{
    Blk blk;
    merge_body<kMergeThreads>(g.a[group_locate(g, &blk)], blk);
}

// Decoder AND merge of a small launch as ONE launch (latency path: B = 1 .. a few dozen images, every workgroup on a CU of its
// own): blocks [0, ndec * B) are the split-stream decoder's (image-major, `ndec` per image), the others the merge bands, which
// do their symbol-independent half while the decoder runs and pick the symbols up through a per-image ticket.  Decoder
// workgroups come first in dispatch order and never wait for a merge workgroup: no deadlock whatever is resident.
struct DecodeMergeArgs {
    DecodeArgs c;
    MergeArgs m;
    unsigned int *done;        // [B] ticket words, kTicketStride apart
    unsigned int ndec, nbands, active_bands, B;
};
__device__ __forceinline__ void decode_merge_body(const DecodeMergeArgs &p, const unsigned int id)
{
    const unsigned int ndec_all = p.ndec * p.B;
    if (id < ndec_all) {
        const unsigned int b = id / p.ndec;
        decode_split_body(p.c, Blk{id - b * p.ndec, b, 0u, p.ndec, p.B});
        __syncthreads();                       // every wave's stores are out (workgroup-scope release)
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_fetch_add(p.done + (size_t)b * kTicketStride, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const unsigned int r = id - ndec_all, b = r / p.nbands;
    merge_body<kDecThreads, true>(p.m, Blk{r - b * p.nbands, b, 0u, p.nbands, p.B}, p.done + (size_t)b * kTicketStride, p.ndec,
                                  p.ndec + p.active_bands);
}
__global__ __launch_bounds__(kDecThreads) CGIC_VGPR_CAP_DECODE void decode_merge_kernel(DecodeMergeArgs p)
{
    decode_merge_body(p, blockIdx.x);
}
__global__ __launch_bounds__(kDecThreads) CGIC_VGPR_CAP_DECODE void decode_merge_grouped_kernel(Grouped<DecodeMergeArgs> g)
{
    Blk blk;
    const DecodeMergeArgs &p = g.a[group_locate(g, &blk)];
    decode_merge_body(p, blk.x);
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

}  // namespace cgic

using namespace cgic;

extern "C" int cgic_decode_stream(const cgic_table *t, const uint8_t *in, int64_t nbytes, int64_t *syms,
                                  int64_t cap, int64_t *count, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_decode_stream");
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

#ifndef CGIC_DEC_WGS_SMALL
#define CGIC_DEC_WGS_SMALL 4
#endif
#ifndef CGIC_DEC_WGS_LARGE
#define CGIC_DEC_WGS_LARGE 24
#endif
#ifdef CGIC_DEV_KNOBS
static int dev_knob_dec(const char *name) { const char *v = getenv(name); return v ? atoi(v) : 0; }
#else
static int dev_knob_dec(const char *) { return 0; }      // the environment knobs exist in `make dbg` builds only
#endif
static int device_cu_count_dec(int *out)
{
    static std::atomic<int> cached{0};
    int n = cached.load();
    if (n == 0) {
        int dev = 0;
        CGIC_HIP_TRY(hipGetDevice(&dev));
        CGIC_HIP_TRY(hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev));
        if (n <= 0) n = 256;
        cached.store(n);
    }
    *out = n;
    return CGIC_OK;
}
static std::atomic<int> g_decode_mode{CGIC_DECODE_AUTO};
static std::atomic<unsigned int *> g_decode_stats{nullptr};
extern "C" int cgic_decode_stats(unsigned int *device_counters)
{
    CGIC_REQUIRE(((uintptr_t)device_counters & 3u) == 0, CGIC_ERR_INVALID, "decode_stats: the counters must be 4-byte aligned");
    g_decode_stats.store(device_counters, std::memory_order_relaxed);
    return CGIC_OK;
}
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
    // 64 bits take the one-wave path of decode_streams_kernel; batches beyond one ticket request are cut into several launches.
    const bool large = h * w > 64 * 64;
    d.bf = (uint32_t *)((char *)d.dcount + align16((size_t)B * 3 * sizeof(int32_t)));
    size_t lds_d = sizeof(uint32_t) * (kDecLutMax + kDecWaves * kSegWinWords) + sizeof(SegShared) + sizeof(int32_t) * 2 * kLdsTrieNodes
                   + sizeof(FastTables);
    if (lds_d < sizeof(uint32_t) * (kDecLutMax + kWinWords)) lds_d = sizeof(uint32_t) * (kDecLutMax + kWinWords);
    if (lds_d > 48 * 1024)
        { int rc_ = ensure_dynamic_lds((const void *)decode_streams_kernel, (size_t)lds_d); if (rc_) return rc_; }
    d.tick = nullptr;
    d.stats = g_decode_stats.load(std::memory_order_relaxed);
    // The self-synchronising one-workgroup-per-image decoder when the worst case of the grid fits its LDS: bits <= symbols
    // the three grids can hold x the longest code.  (Longer inputs are an overflow on any path.)
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
    // Decoder and merge go out as ONE launch when every workgroup of both gets a CU of its own (B = 1 .. a few dozen images, or
    // a few tiles; inside a launch group: within the group's share of the chip)
#ifndef CGIC_DEC_WGS_FUSED_LARGE
#define CGIC_DEC_WGS_FUSED_LARGE 16      // decoder workgroups per large grid when the merge rides in the launch: as fast as 24 per call, and the
                                         // CUs it leaves go to merge bands (2040x1356 chain 0.105 -> 0.104 ms; 12 falls off the fast path)
#endif
    const unsigned int ndec = large ? CGIC_DEC_WGS_FUSED_LARGE : CGIC_DEC_WGS_SMALL;
    int cus = 0;
    rc = device_cu_count_dec(&cus);
    if (rc) return rc;
    const int64_t cu_share = (int64_t)((double)cus * group_cu_share() + 0.5);
    // The fused launch's merge bands SPIN on the decoder workgroups of the same launch (wait_decoded).  Alone on the chip that cannot
    // hang: the decoders sit in front of the bands in the grid and every one gets a CU at once.  With other launches of this kind in
    // flight on other queues (up to four hardware queues by default), an XCD could in principle fill up with spinning bands of
    // several launches whose decoders are queued behind each other's bands.  The chip holds two of these 1024-thread workgroups per
    // CU: as long as FOUR such launches together fit (each at most half the CUs' worth of workgroups), every workgroup of every one
    // of them is resident at once and nobody waits for a slot.  CGIC_DECODE_LATENCY is the caller's statement that this call has
    // the GPU to itself (one batch at a time): it keeps the whole chip as its budget.
    const int64_t cu_budget = dec_mode == CGIC_DECODE_LATENCY ? cu_share : cu_share / 2;
    const bool fuse_base = dec_mode != CGIC_DECODE_THROUGHPUT && d.tab.max_len <= 64 && B * 3 <= (int64_t)(16384 / 4) && !dev_knob_dec("CGIC_NO_DECODE_MERGE");
    int64_t nbands = kMergeBands;
    {
        const int64_t h4 = h >> 2;
#ifndef CGIC_MERGE_MINROWS
#define CGIC_MERGE_MINROWS 1
#endif
#ifndef CGIC_MERGE_WGS
#define CGIC_MERGE_WGS 256
#endif
        while (nbands * B < CGIC_MERGE_WGS && nbands * 2 <= h4 / CGIC_MERGE_MINROWS) {
            // (keep a small launch fusable with its decoder: see below)
            if (fuse_base && B * (int64_t)(ndec + 2 * nbands) > cu_budget && B * (int64_t)(ndec + nbands) <= cu_budget) break;
            nbands *= 2;
        }
    }
    // the image's symbols do not fit LDS: every band stages its own three rank ranges (at most 21/16 symbols per position)
    m.band_syms = 0;
    if (!m.stage_sym) {
        const int64_t rows_per = (((h >> 2) + nbands - 1) / nbands) * 4;
        const int64_t need = rows_per * w * 21 / 16 + 8;
        if (lds_m + (size_t)need * 2 <= 64 * 1024) { m.band_syms = need; lds_m += (size_t)need * 2; }
    }
    // ---- decoder and merge as ONE launch ----
    {
        const int64_t rows_per = (((h >> 2) + nbands - 1) / nbands) * 4;
        const unsigned int active = (unsigned int)((h + rows_per - 1) / rows_per);
        const size_t lds_f = lds_d > lds_m ? lds_d : lds_m;
        if (fuse_base && B * (int64_t)(ndec + nbands) <= cu_budget) {
            DecodeMergeArgs p;
            p.c = d; p.m = m; p.ndec = ndec; p.nbands = (unsigned int)nbands; p.active_bands = active; p.B = (unsigned int)B;
            rc = acquire_tickets(s, (int)(B * 3), &p.c.tick);
            if (rc) return rc;
            rc = acquire_tickets(s, (int)B, &p.done);
            if (rc) return rc;
            rc = ensure_dynamic_lds((const void *)decode_merge_kernel, lds_f);
            if (rc) return rc;
            const dim3 grid_f((unsigned int)(B * (ndec + nbands)));
            return launch_or_record(KID_DECODE_MERGE, grid_f, dim3(kDecThreads), lds_f, p, s, [=] {
                hipLaunchKernelGGL(decode_merge_kernel, grid_f, dim3(kDecThreads), lds_f, s, p);
                return launch_check("decode_merge_kernel"); });
        }
    }
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
            const int sc_ = (int)stage_cap, cc_ = (int)chunk_cap;
            DecodeImageArgs dia;
            dia.a = d; dia.stage_cap = sc_; dia.chunk_cap = cc_;
            static const bool per_stream = getenv("CGIC_SS_PER_STREAM") && atoi(getenv("CGIC_SS_PER_STREAM")) != 0;      // dev A/B, round 6
            if (per_stream && !group_recording()) {
                { int rc_ = ensure_dynamic_lds((const void *)decode_image_stream_kernel, lds_ss); if (rc_) return rc_; }
                hipLaunchKernelGGL(decode_image_stream_kernel, dim3((unsigned)B, 3u), dim3(T), lds_ss, s, d, sc_, cc_);
                rc = launch_check("decode_image_stream_kernel");
            } else
            rc = launch_or_record(KID_DECODE_IMAGE, dim3((unsigned)B), dim3(T), lds_ss, dia, s, [=] {
                hipLaunchKernelGGL(decode_image_kernel, dim3((unsigned)B), dim3(T), lds_ss, s, d, sc_, cc_);
                return launch_check("decode_image_kernel"); });
        }
    }
#endif
    if (ss) {
    } else if (d.tab.max_len <= 64) {
        if (lds_d > 48 * 1024)
            { int rc_ = ensure_dynamic_lds((const void *)decode_split_kernel, (size_t)lds_d); if (rc_) return rc_; }
        // one ticket request covers 3 slots per image: larger batches are cut into several launches of the same kernel
        const int64_t per_launch = (int64_t)(16384 / 4) / 3;
        for (int64_t b0 = 0; b0 < B && rc == CGIC_OK; b0 += per_launch) {
            const int64_t nb = B - b0 < per_launch ? B - b0 : per_launch;
            DecodeArgs c = d;
            c.in = d.in + b0 * CGIC_NUM_STREAMS * slot;
            c.nbytes = d.nbytes + b0 * CGIC_NUM_STREAMS;
            c.dsym = d.dsym + (size_t)b0 * per;
            c.dcount = d.dcount + b0 * 3;
            c.status = d.status ? d.status + b0 : nullptr;
            c.bf = d.bf + (size_t)b0 * 3 * kDecPartsMax * kWave;
            rc = acquire_tickets(s, (int)(nb * 3), &c.tick);
            if (rc) return rc;
            const dim3 grid_c(large ? CGIC_DEC_WGS_LARGE : CGIC_DEC_WGS_SMALL, (unsigned)nb);
            rc = launch_or_record(KID_DECODE_SPLIT, grid_c, dim3(kDecThreads), lds_d, c, s, [=] {
                hipLaunchKernelGGL(decode_split_kernel, grid_c, dim3(kDecThreads), lds_d, s, c);
                return launch_check("decode_split_kernel"); });
        }
    } else {
        rc = launch_or_record(KID_NONE, dim3((unsigned)B, 3), dim3(kDecThreads), lds_d, d, s, [=] {
            hipLaunchKernelGGL(decode_streams_kernel, dim3((unsigned)B, 3), dim3(kDecThreads), lds_d, s, d);
            return launch_check("decode_streams_kernel"); });
    }
    if (rc) return rc;
    if (dec_mode == CGIC_DECODE_THROUGHPUT && !large && m.stage_sym) {
        // several batches in flight: one band of 1024 threads per image (see merge_kernel)
        if (lds_m > 48 * 1024)
            { int rc_ = ensure_dynamic_lds((const void *)merge_kernel<kMergeOneBandThreads>, (size_t)lds_m); if (rc_) return rc_; }
        return launch_or_record(KID_NONE, dim3(1u, (unsigned)B), dim3(kMergeOneBandThreads), lds_m, m, s, [=] {
            hipLaunchKernelGGL(merge_kernel<kMergeOneBandThreads>, dim3(1u, (unsigned)B), dim3(kMergeOneBandThreads), lds_m, s, m);
            return launch_check("merge_kernel"); });
    }
    if (lds_m > 48 * 1024)
        { int rc_ = ensure_dynamic_lds((const void *)merge_kernel<kMergeThreads>, (size_t)lds_m); if (rc_) return rc_; }
    const dim3 grid_m((unsigned)nbands, (unsigned)B);
    return launch_or_record(KID_MERGE, grid_m, dim3(kMergeThreads), lds_m, m, s, [=] {
        hipLaunchKernelGGL(merge_kernel<kMergeThreads>, grid_m, dim3(kMergeThreads), lds_m, s, m);
        return launch_check("merge_kernel"); });
}

static int decode_split_grouped_launch(const GroupRec *const *recs, int n, hipStream_t s)
{
    Grouped<DecodeArgs> g;
    size_t lds;
    int rc = fill_grouped(recs, n, &g, &lds);
    if (rc) return rc;
    if (lds > 48 * 1024) { rc = ensure_dynamic_lds((const void *)decode_split_grouped_kernel, lds); if (rc) return rc; }
    hipLaunchKernelGGL(decode_split_grouped_kernel, dim3(g.start[kMaxGroups]), dim3(kDecThreads), lds, s, g);
    return launch_check("decode_split_grouped_kernel");
}
static int merge_grouped_launch(const GroupRec *const *recs, int n, hipStream_t s)
{
    Grouped<MergeArgs> g;
    size_t lds;
    int rc = fill_grouped(recs, n, &g, &lds);
    if (rc) return rc;
    if (lds > 48 * 1024) { rc = ensure_dynamic_lds((const void *)merge_grouped_kernel, lds); if (rc) return rc; }
    hipLaunchKernelGGL(merge_grouped_kernel, dim3(g.start[kMaxGroups]), dim3(kMergeThreads), lds, s, g);
    return launch_check("merge_grouped_kernel");
}
static int decode_merge_grouped_launch(const GroupRec *const *recs, int n, hipStream_t s)
{
    Grouped<DecodeMergeArgs> g;
    size_t lds;
    int rc = fill_grouped(recs, n, &g, &lds);
    if (rc) return rc;
    rc = ensure_dynamic_lds((const void *)decode_merge_grouped_kernel, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(decode_merge_grouped_kernel, dim3(g.start[kMaxGroups]), dim3(kDecThreads), lds, s, g);
    return launch_check("decode_merge_grouped_kernel");
}
static GroupedRegistrar reg_decode_merge(KID_DECODE_MERGE, decode_merge_grouped_launch);
static GroupedRegistrar reg_decode_split(KID_DECODE_SPLIT, decode_split_grouped_launch);
static GroupedRegistrar reg_merge(KID_MERGE, merge_grouped_launch);

extern "C" int cgic_embedding_gather_f32(const int64_t *ind, int64_t B, int64_t hw, const float *codebook, int K,
                                         int e_dim, float *out, int32_t *status, cgic_stream_t stream)
{
    CGIC_NOT_IN_GROUP("cgic_embedding_gather_f32");
    CGIC_REQUIRE(ind && codebook && out, CGIC_ERR_INVALID, "embedding_gather: NULL argument");
    CGIC_REQUIRE(e_dim == 4 && K > 0, CGIC_ERR_UNSUPPORTED, "embedding_gather: needs a [K,4] codebook");
    const int64_t n = B * hw;
    if (n <= 0) return CGIC_OK;
    int nblk = (int)((n + 255) / 256);
    if (nblk > 4096) nblk = 4096;
    hipLaunchKernelGGL(gather_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, ind, B, hw, codebook, K, out, status);
    return launch_check("gather_kernel");
}
