// gpu_entropy.go — goes into github.com/flanglet/kanzi-go/v2/entropy.
//
// kanzi.EntropyEncoder / kanzi.EntropyDecoder (v2/Definitions.go:154-179) backed by knz_entropy_encode /
// knz_entropy_decode of libknz_gpu.so for HUFFMAN, ANS0, ANS1, FPAQ and NONE: what entropy.NewEntropyEncoder /
// NewEntropyDecoder (EntropyCodecFactory.go:45-134) hand out for those types, for callers that hold a device handle and
// build single codec objects. The factories themselves are not patched: the stream does not go through these objects
// (its boundary is the block batch, gpu_stream.go); the reference's own unit tests reach them through the helpers of
// testhooks/gpu_hooks_entropy_test.go.
package entropy

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

	kanzi "github.com/flanglet/kanzi-go/v2"
)

// GPUEntropyEncoder encodes a block on the device and appends the resulting bit string to the bitstream.
type GPUEntropyEncoder struct {
	h   unsafe.Pointer
	typ uint32
	bs  kanzi.OutputBitStream
	buf []byte
}

// NewGPUEntropyEncoder wraps a handle opened by the stream for one entropy codec type (HUFFMAN_TYPE, ANS0_TYPE, ...).
func NewGPUEntropyEncoder(handle unsafe.Pointer, bs kanzi.OutputBitStream, entropyType uint32) (*GPUEntropyEncoder, error) {
	if handle == nil || bs == nil {
		return nil, errors.New("GPU entropy encoder: missing handle or bitstream")
	}

	if C.knz_supports(0, C.uint32_t(entropyType)) != 1 {
		return nil, fmt.Errorf("GPU entropy encoder: codec type %d has no device implementation", entropyType)
	}

	return &GPUEntropyEncoder{h: handle, typ: entropyType, bs: bs}, nil
}

// Write encodes the data provided into the bitstream. Return the number of bytes written to the bitstream.
// The bits appended are identical to what HuffmanEncoder / ANSRangeEncoder / FPAQEncoder.Write emit for the block.
func (this *GPUEntropyEncoder) Write(block []byte) (int, error) {
	if block == nil {
		return 0, errors.New("Invalid null block parameter")
	}

	if len(block) == 0 {
		return 0, nil
	}

	if need := 2*len(block) + 262144; len(this.buf) < need {
		this.buf = make([]byte, need)
	}

	var bits C.uint64_t
	rc := C.knz_entropy_encode(this.h, C.uint32_t(this.typ), (*C.uint8_t)(unsafe.Pointer(&block[0])), C.uint32_t(len(block)),
		(*C.uint8_t)(unsafe.Pointer(&this.buf[0])), C.uint64_t(len(this.buf)), &bits)

	if rc != 0 {
		return 0, fmt.Errorf("GPU entropy encoder: %s (error %d)", C.GoString(C.knz_last_error(this.h)), int(rc))
	}

	const chunk = uint64(1) << 30

	for ofs, rest := uint64(0), uint64(bits); rest > 0; {
		sz := rest

		if sz > chunk {
			sz = chunk
		}

		this.bs.WriteArray(this.buf[ofs>>3:], uint(sz))
		ofs += sz
		rest -= sz
	}

	return len(block), nil
}

// BitStream returns the underlying bitstream
func (this *GPUEntropyEncoder) BitStream() kanzi.OutputBitStream {
	return this.bs
}

// Dispose must be called before getting rid of the entropy encoder (nothing is pending: every Write is complete)
func (this *GPUEntropyEncoder) Dispose() {
}

// GPUEntropyDecoder decodes blocks on the device. kanzi.InputBitStream has no "remaining bytes" call
// (Definitions.go:94-116), so the decoder owns the payload: inside the stream reader it is data[ofs:r] of
// decodingTask.decode (CompressedStream.go:1875-1914), ofs = ibs.Read()/8 behind the block header fields
// (mode, skip flags, length, checksum: whole bytes, so the entropy payload starts on a byte).
type GPUEntropyDecoder struct {
	h       unsafe.Pointer
	typ     uint32
	bs      kanzi.InputBitStream
	payload []byte
	bitPos  uint64
}

// NewGPUEntropyDecoder wraps a handle for one codec type over a payload held by the caller. bs may be nil; when it is
// given, Read also consumes the decoded bits from it so that bs.Read() stays what the Go decoder would leave.
func NewGPUEntropyDecoder(handle unsafe.Pointer, bs kanzi.InputBitStream, payload []byte, entropyType uint32) (*GPUEntropyDecoder, error) {
	if handle == nil {
		return nil, errors.New("GPU entropy decoder: no device handle")
	}

	if C.knz_supports(0, C.uint32_t(entropyType)) != 1 {
		return nil, fmt.Errorf("GPU entropy decoder: codec type %d has no device implementation", entropyType)
	}

	return &GPUEntropyDecoder{h: handle, typ: entropyType, bs: bs, payload: payload}, nil
}

// Read decodes data from the bitstream and return it in the provided buffer. Return the number of bytes read from
// the bitstream.
func (this *GPUEntropyDecoder) Read(block []byte) (int, error) {
	if block == nil {
		return 0, errors.New("Invalid null block parameter")
	}

	if len(block) == 0 {
		return 0, nil
	}

	if this.bitPos&7 != 0 {
		return 0, errors.New("GPU entropy decoder: payload position is not on a byte")
	}

	rest := this.payload[this.bitPos>>3:]

	if len(rest) == 0 {
		return 0, errors.New("GPU entropy decoder: no payload left")
	}

	var used C.uint64_t
	rc := C.knz_entropy_decode(this.h, C.uint32_t(this.typ), (*C.uint8_t)(unsafe.Pointer(&rest[0])), C.uint64_t(len(rest)),
		(*C.uint8_t)(unsafe.Pointer(&block[0])), C.uint32_t(len(block)), &used)

	if rc != 0 {
		return 0, fmt.Errorf("GPU entropy decoder: %s (error %d)", C.GoString(C.knz_last_error(this.h)), int(rc))
	}

	this.bitPos += uint64(used)

	if this.bs != nil { // keep the shared bitstream where the Go decoder would leave it
		for n := uint64(used); n > 0; {
			k := uint(64)

			if n < 64 {
				k = uint(n)
			}

			this.bs.ReadBits(k)
			n -= uint64(k)
		}
	}

	return len(block), nil
}

// BitStream returns the underlying bitstream (nil when the decoder was built over a bare payload)
func (this *GPUEntropyDecoder) BitStream() kanzi.InputBitStream {
	return this.bs
}

// Dispose must be called before getting rid of the entropy decoder
func (this *GPUEntropyDecoder) Dispose() {
}
