// cgic_coder_dev.h -- what the three translation units of the entropy coder share (cgic_coder.hip: encode side,
// cgic_decode.hip: prefix decoders of the latency mode + scatter / merge / gather, cgic_decode_ss.hip: the
// self-synchronising decoder of the throughput mode).
#pragma once
#include "cgic_common.h"

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

constexpr int kDecLutMax = 1 << kLutBitsMax;        // 13-bit LUT

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
#ifndef CGIC_DEC_PARTS_MAX
#define CGIC_DEC_PARTS_MAX 12
#endif
constexpr int kDecPartsMax = CGIC_DEC_PARTS_MAX;                    // ... and at most in the one-launch form: flags 0..11 and the reader count (word 15) share one ticket slot
constexpr int kDecDoneWord = 15;
#ifndef CGIC_DEC_PART_BYTES
#define CGIC_DEC_PART_BYTES 1280
#endif
constexpr int kDecPartBytes = CGIC_DEC_PART_BYTES;     // stream bytes per part: 160 chunks, one pass-A round of 16 waves x 10 chunks

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
    uint32_t *bf;            // [B, 3, parts, 64] range functions exchanged by the parts of decode_split_kernel
    unsigned int *tick;      // [B, 3] ticket slots of decode_split_kernel
    unsigned int *stats;     // telemetry of the self-synchronising decoder (cgic_decode_stats) or NULL: [0] sweeps summed over the
                             // images, [1] images, [2] most sweeps of one image
};

// the self-synchronising one-workgroup-per-image decoder (cgic_decode_ss.hip); launched by cgic_decompress_streams
#ifndef CGIC_SS_THREADS_SMALL
#define CGIC_SS_THREADS_SMALL 256
#endif
__global__ void decode_image_kernel(DecodeArgs a, int stage_cap, int chunk_cap);
__global__ void decode_image_stream_kernel(DecodeArgs a, int stage_cap, int chunk_cap);
struct DecodeImageArgs { DecodeArgs a; int stage_cap, chunk_cap; };      // its argument block in a grouped launch

static const int kModeStreams[7] = {0x1f, 0x16, 0x0d, 0x0b, 0x01, 0x02, 0x04};  // model.py:225-260

static inline size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

static inline int check_grid(int64_t B, int64_t h, int64_t w, int mode)
{
    CGIC_REQUIRE(B >= 0 && h > 0 && w > 0 && h % 4 == 0 && w % 4 == 0, CGIC_ERR_INVALID,
                 "latent grid %lldx%lld must be positive multiples of 4", (long long)h, (long long)w);
    CGIC_REQUIRE(mode >= 0 && mode <= 6, CGIC_ERR_INVALID, "mode %d outside 0..6", mode);
    CGIC_REQUIRE(B <= 65535, CGIC_ERR_UNSUPPORTED, "batch %lld exceeds the grid limit", (long long)B);
    CGIC_REQUIRE(h * w < ((int64_t)1 << 26), CGIC_ERR_UNSUPPORTED, "latent grid too large");
    return CGIC_OK;
}

}  // namespace cgic
