// `-s` (ctx["skipBlocks"], v2/io/CompressedStream.go:778-800): a block whose first bytes are the magic number of a compressed
// format (internal/Magic.go:83-170) or whose order-0 entropy is >= 973/1024 of 8 bits per byte (internal/Global.go:174-216,
// entropy/EntropyUtils.go:26) is emitted as a copy block (mode bit 0x80, no transform, raw bytes). One workgroup per block
// decides; the entropy stage of the batch then runs as usual and the chunks of copy blocks are overwritten with their raw
// bytes (knz_copy_units_kernel) - incompressible blocks are the exception, a second code path through five codecs is not.
#include "bits.h"

// 4096 * log2(x) rounded to nearest, x = 0..256 (Global.go:59-88 holds the same numbers; tests compare the two)
__device__ const uint16_t KNZ_LOG2_4096[257] = {
    0, 0, 4096, 6492, 8192, 9511, 10588, 11499, 12288, 12984, 13607, 14170,
    14684, 15157, 15595, 16003, 16384, 16742, 17080, 17400, 17703, 17991, 18266, 18529,
    18780, 19021, 19253, 19476, 19691, 19898, 20099, 20292, 20480, 20662, 20838, 21010,
    21176, 21338, 21496, 21649, 21799, 21945, 22087, 22226, 22362, 22495, 22625, 22752,
    22876, 22998, 23117, 23234, 23349, 23462, 23572, 23680, 23787, 23892, 23994, 24095,
    24195, 24292, 24388, 24483, 24576, 24668, 24758, 24847, 24934, 25021, 25106, 25189,
    25272, 25354, 25434, 25513, 25592, 25669, 25745, 25820, 25895, 25968, 26041, 26112,
    26183, 26253, 26322, 26390, 26458, 26525, 26591, 26656, 26721, 26784, 26848, 26910,
    26972, 27033, 27094, 27154, 27213, 27272, 27330, 27388, 27445, 27502, 27558, 27613,
    27668, 27722, 27776, 27830, 27883, 27935, 27988, 28039, 28090, 28141, 28191, 28241,
    28291, 28340, 28388, 28437, 28484, 28532, 28579, 28626, 28672, 28718, 28764, 28809,
    28854, 28898, 28943, 28987, 29030, 29074, 29117, 29159, 29202, 29244, 29285, 29327,
    29368, 29409, 29450, 29490, 29530, 29570, 29609, 29649, 29688, 29726, 29765, 29803,
    29841, 29879, 29916, 29954, 29991, 30027, 30064, 30100, 30137, 30172, 30208, 30244,
    30279, 30314, 30349, 30384, 30418, 30452, 30486, 30520, 30554, 30587, 30621, 30654,
    30687, 30719, 30752, 30784, 30817, 30849, 30880, 30912, 30944, 30975, 31006, 31037,
    31068, 31099, 31129, 31160, 31190, 31220, 31250, 31280, 31309, 31339, 31368, 31397,
    31426, 31455, 31484, 31513, 31541, 31569, 31598, 31626, 31654, 31681, 31709, 31737,
    31764, 31791, 31818, 31846, 31872, 31899, 31926, 31952, 31979, 32005, 32031, 32058,
    32084, 32109, 32135, 32161, 32186, 32212, 32237, 32262, 32287, 32312, 32337, 32362,
    32387, 32411, 32436, 32460, 32484, 32508, 32533, 32557, 32580, 32604, 32628, 32651,
    32675, 32698, 32722, 32745, 32768,
};

__device__ __forceinline__ uint32_t knz_log2_scaled_1024(uint32_t x) {          // Log2ScaledBy1024, x >= 1
    if (x < 256) return ((uint32_t)KNZ_LOG2_4096[x] + 2) >> 2;
    const uint32_t lg = 31u - (uint32_t)__builtin_clz(x);
    if ((x & (x - 1)) == 0) return lg << 10;
    return ((lg - 7) * 1024) + (((uint32_t)KNZ_LOG2_4096[x >> (lg - 7)] + 2) >> 2);
}

__device__ __forceinline__ bool knz_magic_is_compressed(uint32_t key) {           // IsDataCompressed(GetMagicType(.))
    if (key == 0xFFD8FFE0u) return true;                                         // JPG: the switch only lists the E0 form
    const uint32_t k24 = key >> 8, k16 = key >> 16;
    if ((key & ~0x0Fu) == 0xFFD8FFE0u) return false;                             // other JPG markers: recognised first, not in the list
    if (k24 == 0x425A68u || k24 == 0x494433u) return true;                       // BZIP2, MP3 ID3
    switch (key) {
        case 0x47494638u: case 0x89504E47u: case 0x377ABCAFu: case 0x28B52FFDu: case 0x81CFB2CEu: case 0x4D534346u: case 0x504B0304u:
        case 0x664C6143u: case 0xFD377A58u: case 0x4B414E5Au: case 0x52617221u:
            return true;                                                         // GIF PNG LZMA ZSTD BROTLI CAB ZIP FLAC XZ KNZ RAR
        // 32-bit magics that are recognised but not "compressed" end the search before the 16-bit ones (GetMagicType returns them)
        case 0x25504446u: case 0x7F454C46u: case 0xFEEDFACEu: case 0xCEFAEDFEu: case 0xFEEDFACFu: case 0xCFFAEDFEu: case 0x52494646u:
            return false;
        default: break;
    }
    return k16 == 0x1F8Bu;                                                       // GZIP (BMP, WIN, PBM.. are not compressed types)
}

struct SkipArgs {
    uint32_t nblocks;
    const uint64_t* blk_off; const uint32_t* blk_len;
    uint8_t* blk_copy; uint8_t* blk_skip; uint8_t* active;      // active may be null (no transform stage)
};

__global__ __launch_bounds__(256) void knz_skip_detect_kernel(SkipArgs a) {
    __shared__ uint32_t s_hist[4][256];
    __shared__ unsigned long long s_sum;
    const int tid = threadIdx.x, wave = tid >> 6;
    const uint32_t b = blockIdx.x;
    const uint32_t n = a.blk_len[b];
    if (n <= 15) return;                                                          // copy block already (:773-776)
    const uint8_t* src = (const uint8_t*)a.blk_off[b];
    bool skip = false;
    if (n >= 8) skip = knz_magic_is_compressed(((uint32_t)src[0] << 24) | ((uint32_t)src[1] << 16) | ((uint32_t)src[2] << 8) | src[3]);
    if (!skip) {
        for (int i = tid; i < 4 * 256; i += 256) (&s_hist[0][0])[i] = 0;
        if (tid == 0) s_sum = 0;
        __syncthreads();
        const uint32_t nw = (((uintptr_t)src) & 3) == 0 ? n >> 2 : 0;            // whole words when the block is aligned
        for (uint32_t i = tid; i < nw; i += 256) {
            const uint32_t w = ((const uint32_t*)src)[i];
            atomicAdd(&s_hist[wave][w & 0xFF], 1u); atomicAdd(&s_hist[wave][(w >> 8) & 0xFF], 1u);
            atomicAdd(&s_hist[wave][(w >> 16) & 0xFF], 1u); atomicAdd(&s_hist[wave][w >> 24], 1u);
        }
        for (uint32_t i = 4 * nw + tid; i < n; i += 256) atomicAdd(&s_hist[wave][src[i]], 1u);
        __syncthreads();
        const uint32_t f = s_hist[0][tid] + s_hist[1][tid] + s_hist[2][tid] + s_hist[3][tid];
        if (f) atomicAdd(&s_sum, ((unsigned long long)f * (unsigned long long)(knz_log2_scaled_1024(n) - knz_log2_scaled_1024(f))) >> 3);
        __syncthreads();
        skip = (int)(s_sum / (unsigned long long)n) >= 973;
    }
    if (skip && tid == 0) {
        a.blk_copy[b] = 1;
        a.blk_skip[b] = 0x7F;                                                     // a lone NullTransform that applied
        if (a.active) a.active[b] = 0;
    }
}

struct CopyUnitsArgs {
    uint32_t chunks_per_block, chunk_size;
    const uint8_t* blk_copy; const uint64_t* blk_off; const uint32_t* blk_len;
    uint8_t* scratch; uint64_t slot_stride;
    uint32_t* unit_bits; uint32_t* unit_src; int32_t* blk_status;
};

// chunks of (large) copy blocks: one unit of raw bytes at the start of the slot, whatever the entropy stage left there
__global__ __launch_bounds__(256) void knz_copy_units_kernel(CopyUnitsArgs a) {
    const int tid = threadIdx.x;
    const uint32_t b = blockIdx.x / a.chunks_per_block, k = blockIdx.x % a.chunks_per_block;
    const uint32_t len = a.blk_len[b];
    if (!a.blk_copy[b] || len <= 15) return;                                      // small copy blocks are raw in every codec already
    uint32_t* ubits = a.unit_bits + (size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK;
    uint32_t* usrc = a.unit_src + (size_t)blockIdx.x * KNZ_UNITS_PER_CHUNK;
    const uint64_t lo = (uint64_t)k * a.chunk_size;
    const uint32_t n = lo < len ? (uint32_t)min((uint64_t)a.chunk_size, len - lo) : 0u;
    if (tid < KNZ_UNITS_PER_CHUNK) { ubits[tid] = tid == 0 ? 8u * n : 0u; if (tid == 0) usrc[0] = 0; }
    if (k == 0 && tid == 0) a.blk_status[b] = 0;                                  // nothing the skipped codec complained about applies
    if (n == 0) return;
    const uint8_t* src = (const uint8_t*)a.blk_off[b] + lo;
    uint8_t* dst = a.scratch + (size_t)blockIdx.x * a.slot_stride;
    if (((((uintptr_t)src) | ((uintptr_t)dst)) & 15) == 0) {
        for (uint32_t i = tid; i < (n >> 4); i += 256) ((uint4*)dst)[i] = ((const uint4*)src)[i];
        for (uint32_t i = (n & ~15u) + tid; i < n; i += 256) dst[i] = src[i];
    } else {
        for (uint32_t i = tid; i < n; i += 256) dst[i] = src[i];
    }
}
