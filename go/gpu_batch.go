// gpu_batch.go — goes into github.com/flanglet/kanzi-go/v2/io next to CompressedStream.go.
//
// Re-points the per-block goroutine fan-out of Writer.processBlock (CompressedStream.go:636-710) and
// Reader.processBlock (:1633-1744) at the GPU batch scheduler of libknz_gpu.so (include/knz_gpu.h).
// The ordered emission (:934-976), the sequential payload reads (:1816-1852), the listeners and every
// error path stay in Go. No Go toolchain exists in the image this library is built in: this file is the
// source a kanzi-go maintainer adds; the same C entry points are exercised by the Python mirror
// (kanzi-go_amd/api.py) and by tests/csmoke/smoke.c.
package io

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -lknz_gpu
#include <stdlib.h>
#include "knz_gpu.h"
*/
import "C"

import (
	"runtime"
	"unsafe"

	kanzi "github.com/flanglet/kanzi-go/v2"
	"github.com/flanglet/kanzi-go/v2/internal"
)

// gpuBatch is one GPU batch scheduler: what a Writer or a Reader owns instead of its task pool.
type gpuBatch struct {
	h unsafe.Pointer
}

// newGPUBatch opens the scheduler for the stream's configuration. transformType / entropyType are the values
// transform.GetType and entropy.GetType return (Factory.go:289-328, EntropyCodecFactory.go:173-206).
// skipBlocks is ctx["skipBlocks"] (the CLI's -s). devices: nil = the current HIP device; otherwise the HIP ordinals the
// batches fan out over, one lane each (knz_open_devices: contiguous balanced block ranges, every device copies its own
// blocks in and out, no collective; an ordinal may be named several times: its lanes overlap each other's copies and
// kernels). The library has no CPU fallback: with no usable GPU this fails and the caller keeps the goroutine path.
func newGPUBatch(transformType uint64, entropyType uint32, blockSize int, checksumBits int, skipBlocks bool, devices []int) (*gpuBatch, error) {
	var cfg C.knz_cfg
	cfg.transform = C.uint64_t(transformType)
	cfg.entropy = C.uint32_t(entropyType)
	cfg.block_size = C.uint32_t(blockSize)
	cfg.checksum_bits = C.uint32_t(checksumBits)
	cfg.bs_version = 6
	cfg.device = -1

	if skipBlocks {
		cfg.flags = C.KNZ_FLAG_SKIP_BLOCKS
	}

	b := &gpuBatch{}
	rc := C.int(0)

	if len(devices) == 0 {
		rc = C.knz_open(&cfg, &b.h)
	} else {
		if len(devices) > _MAX_CONCURRENCY {
			return nil, &IOError{msg: "Too many GPU lanes", code: kanzi.ERR_INVALID_PARAM}
		}

		n := len(devices)
		ords := (*[_MAX_CONCURRENCY]C.int32_t)(C.calloc(C.size_t(n), C.size_t(4)))[:n:n]
		defer C.free(unsafe.Pointer(&ords[0]))

		for i := range ords {
			ords[i] = C.int32_t(devices[i])
		}

		rc = C.knz_open_devices(&cfg, &ords[0], C.int(n), &b.h)
	}

	if rc != 0 {
		return nil, &IOError{msg: "Cannot open the GPU batch scheduler: " + C.GoString(C.knz_last_error(nil)), code: int(rc)}
	}

	runtime.SetFinalizer(b, func(x *gpuBatch) { x.close() })
	return b, nil
}

func (b *gpuBatch) close() {
	if b.h != nil {
		C.knz_close(b.h)
		b.h = nil
	}
}

// gpuBlockResult is what one encodingTask has at obs.Close() (CompressedStream.go:912-914) plus the values its
// events carry (:803-806, :829-832, :916-919).
type gpuBlockResult struct {
	written   uint64 // exact bit count of the block-local stream
	postLen   int    // size behind the transform sequence (EVT_AFTER_TRANSFORM)
	skipFlags byte
	mode      byte
	checksum  uint64
}

// encodeBlocks encodes the buffered blocks in one device batch. data[i][:lengths[i]] are the blocks, out[i]
// receives the block-local stream (mode byte .. entropy payload) exactly as encodingTask.encode leaves it.
// C memory is used for the descriptors so that no Go pointer to Go pointers crosses the boundary.
func (b *gpuBatch) encodeBlocks(data [][]byte, lengths []int, out [][]byte) ([]gpuBlockResult, error) {
	n := len(lengths)

	if n == 0 {
		return nil, nil
	}

	blocks := (*[1 << 20]C.knz_block)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.knz_block{}))))[:n:n]
	defer C.free(unsafe.Pointer(&blocks[0]))
	var pinner runtime.Pinner
	defer pinner.Unpin()

	for i := 0; i < n; i++ {
		pinner.Pin(&data[i][0])
		pinner.Pin(&out[i][0])
		blocks[i].src = (*C.uint8_t)(unsafe.Pointer(&data[i][0]))
		blocks[i].src_len = C.uint32_t(lengths[i])
		blocks[i].dst = (*C.uint8_t)(unsafe.Pointer(&out[i][0]))
		blocks[i].dst_cap = C.uint32_t(len(out[i]))
	}

	if rc := C.knz_encode_blocks(b.h, &blocks[0], C.int(n)); rc != 0 {
		return nil, &IOError{msg: C.GoString(C.knz_last_error(b.h)), code: int(rc)} // rc is a kanzi.ERR_* value
	}

	res := make([]gpuBlockResult, n)

	for i := range res {
		res[i] = gpuBlockResult{written: uint64(blocks[i].out_bits), postLen: int(blocks[i].post_len),
			skipFlags: byte(blocks[i].skip_flags), mode: byte(blocks[i].mode), checksum: uint64(blocks[i].checksum)}
	}

	return res, nil
}

// emitBlocks is the tail of Writer.processBlock, unchanged (CompressedStream.go:951-976): ordered emission of the
// block-local streams into the shared bitstream, once per block.
func emitBlocks(obs kanzi.OutputBitStream, out [][]byte, res []gpuBlockResult) {
	for i := range res {
		written := res[i].written
		lw := uint(3)

		if written >= 8 {
			lw = uint(internal.Log2NoCheck(uint32(written>>3)) + 4)
		}

		obs.WriteBits(uint64(lw-3), 5) // write length-3 (5 bits max)
		obs.WriteBits(written, lw)

		// chunked like the reference: WriteArray takes a bit count that fits in a uint
		const chunk = uint64(1) << 30
		ofs := uint64(0)
		rest := written

		for rest > 0 {
			sz := rest

			if sz > chunk {
				sz = chunk
			}

			obs.WriteArray(out[i][ofs>>3:], uint(sz))
			ofs += sz
			rest -= sz
		}
	}
}

// decodeBlocks decodes the payloads the Reader has read from the shared stream (payload[i] = data[0:r] of
// decodingTask.decode, :1816-1852) into out[i]; returns the decoded sizes. Replaces the concurrent bodies
// (:1875-2011). A checksum mismatch comes back as kanzi.ERR_CRC_CHECK, a damaged payload as ERR_PROCESS_BLOCK.
func (b *gpuBatch) decodeBlocks(payload [][]byte, out [][]byte) ([]int, error) {
	n := len(payload)

	if n == 0 {
		return nil, nil
	}

	blocks := (*[1 << 20]C.knz_block)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.knz_block{}))))[:n:n]
	defer C.free(unsafe.Pointer(&blocks[0]))
	var pinner runtime.Pinner
	defer pinner.Unpin()

	for i := 0; i < n; i++ {
		pinner.Pin(&payload[i][0])
		pinner.Pin(&out[i][0])
		blocks[i].src = (*C.uint8_t)(unsafe.Pointer(&payload[i][0]))
		blocks[i].src_len = C.uint32_t(len(payload[i]))
		blocks[i].dst = (*C.uint8_t)(unsafe.Pointer(&out[i][0]))
		blocks[i].dst_cap = C.uint32_t(len(out[i]))
	}

	if rc := C.knz_decode_blocks(b.h, &blocks[0], C.int(n)); rc != 0 {
		return nil, &IOError{msg: C.GoString(C.knz_last_error(b.h)), code: int(rc)}
	}

	sizes := make([]int, n)

	for i := range sizes {
		sizes[i] = int(blocks[i].out_bits)
	}

	return sizes, nil
}

// GPUDeviceCount returns the number of HIP devices the library sees (0 without a usable GPU).
func GPUDeviceCount() int {
	return int(C.knz_device_count())
}

// gpuSupports tells whether a transform / entropy combination has a device implementation in the linked build
// (the host keeps its goroutine path otherwise).
func gpuSupports(transformType uint64, entropyType uint32) bool {
	return C.knz_supports(C.uint64_t(transformType), C.uint32_t(entropyType)) == 1
}
