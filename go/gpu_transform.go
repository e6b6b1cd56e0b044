// gpu_transform.go — goes into github.com/flanglet/kanzi-go/v2/transform.
//
// A kanzi.ByteTransform (v2/Definitions.go:78-91) backed by knz_transform_forward / knz_transform_inverse of
// libknz_gpu.so for the transforms of the hot path: BWT (block codec form), RANK, MTFT, ZRLT, LZ, LZX, LZP, SRT, UTF, TEXT (DICT_TYPE).
// What transform.New (Factory.go:97-185) builds for those ids, for callers that hold a device handle and want single
// transform objects. The factory itself is not patched: the stream does not go through these objects (its boundary is the
// block batch, gpu_stream.go); the reference's own unit tests reach them through testhooks/gpu_hooks_transform_test.go.
package transform

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -lknz_gpu
#include "knz_gpu.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"unsafe"
)

// GPUTransform is one transform object. id is the 6-bit transform id of Factory.go:31-53 (BWT_TYPE, RANK_TYPE, ...).
type GPUTransform struct {
	h  unsafe.Pointer // handle of knz_open, owned by the Writer / Reader
	id uint64
}

// NewGPUTransform wraps a handle opened by the stream (io.gpuBatch) for one transform id.
func NewGPUTransform(handle unsafe.Pointer, id uint64) (*GPUTransform, error) {
	if handle == nil {
		return nil, errors.New("GPU transform: no device handle")
	}

	if C.knz_supports(C.uint64_t(id<<42), 0) != 1 {
		return nil, fmt.Errorf("GPU transform: transform id %d has no device implementation", id)
	}

	return &GPUTransform{h: handle, id: id}, nil
}

// Forward applies the function to the src and writes the result to the destination. Returns number of bytes read,
// number of bytes written and possibly an error. As everywhere in kanzi-go an error from Forward means "skip this
// transform" (Sequence.go:86-91): the library reports a declined transform as KNZ_SKIP.
func (this *GPUTransform) Forward(src, dst []byte) (uint, uint, error) {
	if len(src) == 0 || len(dst) == 0 {
		return 0, 0, nil
	}

	if &src[0] == &dst[0] {
		return 0, 0, errors.New("Input and output buffers cannot be equal")
	}

	var n C.uint32_t
	rc := C.knz_transform_forward(this.h, C.uint64_t(this.id), (*C.uint8_t)(unsafe.Pointer(&src[0])), C.uint32_t(len(src)),
		(*C.uint8_t)(unsafe.Pointer(&dst[0])), C.uint32_t(len(dst)), &n)

	if rc == C.KNZ_SKIP {
		return 0, 0, errors.New("GPU transform: forward transform skipped")
	}

	if rc != 0 {
		return 0, 0, fmt.Errorf("GPU transform: forward failed: %s (error %d)", C.GoString(C.knz_last_error(this.h)), int(rc))
	}

	return uint(len(src)), uint(n), nil
}

// Inverse applies the reverse function to the src and writes the result to the destination. Any error is fatal for
// the block (Sequence.go:160-170).
func (this *GPUTransform) Inverse(src, dst []byte) (uint, uint, error) {
	if len(src) == 0 || len(dst) == 0 {
		return 0, 0, nil
	}

	if &src[0] == &dst[0] {
		return 0, 0, errors.New("Input and output buffers cannot be equal")
	}

	var n C.uint32_t
	rc := C.knz_transform_inverse(this.h, C.uint64_t(this.id), (*C.uint8_t)(unsafe.Pointer(&src[0])), C.uint32_t(len(src)),
		(*C.uint8_t)(unsafe.Pointer(&dst[0])), C.uint32_t(len(dst)), &n)

	if rc != 0 {
		return 0, 0, fmt.Errorf("GPU transform: inverse failed: %s (error %d)", C.GoString(C.knz_last_error(this.h)), int(rc))
	}

	return uint(len(src)), uint(n), nil
}

// MaxEncodedLen returns the max size required for the encoding output buffer (the same bound the Go transform gives).
func (this *GPUTransform) MaxEncodedLen(srcLen int) int {
	return int(C.knz_max_encoded_len(C.uint64_t(this.id<<42), C.uint32_t(srcLen)))
}
