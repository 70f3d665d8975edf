"""one-off source edit (kept for the record): size-based part counts + prologue order of decode_split_kernel"""
import sys
p = sys.argv[1]
s = open(p).read()
a_ = s.index("// (scalars only, no indexed arrays: everything here stays on the scalar unit)")
b_ = s.index("__global__ __launch_bounds__(kDecThreads) void decode_split_kernel(DecodeArgs a)")
roles = '''// (scalars only, no indexed arrays, no loop in the common case: everything here stays on the scalar unit)
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

'''
s = s[:a_] + roles + s[b_:]

old_a = s.index("    // who am I: every workgroup of the image derives the same split from the same three byte counts")
old_b = s.index("    uint16_t *dst = a.dsym + b * (n_c + n_m + n_f) + off;\n    auto put = [&](int k, int sym) { dst[k] = (uint16_t)sym; };\n    FastTables *ft = reinterpret_cast<FastTables *>(reinterpret_cast<int32_t *>(seg + 1) + 2 * kLdsTrieNodes);\n    if (nparts == 1) {")
new = '''    // who am I: every workgroup of the image derives the same split from the same three byte counts.  The counts and the
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
'''
s = s[:old_a] + new + s[old_b:]
s = s.replace("    __shared__ int s_count, s_nb, s_pad;\n    __shared__ uint32_t s_fn[kWave];\n    __shared__ int s_entry[2];\n    uint32_t *lut = sm;\n    uint32_t *win = lut + kDecLutMax;\n    SegShared *seg = reinterpret_cast<SegShared *>(win + kDecWaves * kSegWinWords);\n    const int tid = threadIdx.x, wave = tid >> 6;\n    const int64_t b = blockIdx.y;\n    // who am I",
              "    __shared__ int s_count;\n    __shared__ uint32_t s_fn[kWave];\n    __shared__ int s_entry[2];\n    uint32_t *lut = sm;\n    uint32_t *win = lut + kDecLutMax;\n    SegShared *seg = reinterpret_cast<SegShared *>(win + kDecWaves * kSegWinWords);\n    const int tid = threadIdx.x, wave = tid >> 6;\n    const int64_t b = blockIdx.y;\n    // who am I", 1)
s = s.replace("constexpr int kDecMinPartBytes = 128;               // a part smaller than 16 chunks (one per wave) is not worth a workgroup",
              "#ifndef CGIC_DEC_PART_BYTES\n#define CGIC_DEC_PART_BYTES 1280\n#endif\nconstexpr int kDecPartBytes = CGIC_DEC_PART_BYTES;     // stream bytes per part: 160 chunks, one pass-A round of 16 waves x 10 chunks", 1)
open(p, 'w').write(s)
print("ok")
