// gpu_hooks_entropy_test.go — goes into github.com/flanglet/kanzi-go/v2/entropy next to Entropy_test.go (test build only).
//
// Entropy_test.go builds every codec it tests through two helpers, getEncoder / getDecoder (Entropy_test.go:560-588). With
// the one-line hook at the top of each (tools/go2cpp/apply_test_patch.py; shown in INTEGRATION.md) the helpers hand out
// the device-backed objects of gpu_entropy.go for the codecs the library implements, and the reference's own tests
// (TestHuffman, TestANS0, TestANS1, TestFPAQ, TestFPAQCodecSpecificPatterns) run against the device unchanged.
package entropy

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -lknz_gpu
#include "knz_gpu.h"
*/
import "C"

import (
	"unsafe"

	kanzi "github.com/flanglet/kanzi-go/v2"
)

var gpuTestHandle unsafe.Pointer
var gpuTestObjects int

// GpuTestObjects is the number of device-backed codec objects the tests were handed so far.
func GpuTestObjects() int {
	return gpuTestObjects
}

func gpuTestOpen() unsafe.Pointer {
	if gpuTestHandle == nil {
		var cfg C.knz_cfg
		cfg.block_size = C.uint32_t(4 << 20)
		cfg.bs_version = 6
		cfg.device = -1

		if rc := C.knz_open(&cfg, &gpuTestHandle); rc != 0 {
			panic("Cannot open the device: " + C.GoString(C.knz_last_error(nil)))
		}
	}

	return gpuTestHandle
}

func gpuTestEncoder(name string, obs kanzi.OutputBitStream) kanzi.EntropyEncoder {
	eType, err := GetType(name)

	if err != nil || C.knz_supports(0, C.uint32_t(eType)) != 1 {
		return nil
	}

	res, err := NewGPUEntropyEncoder(gpuTestOpen(), obs, eType)

	if err != nil {
		panic(err.Error())
	}

	gpuTestObjects++
	return res
}

func gpuTestDecoder(name string, ibs kanzi.InputBitStream) kanzi.EntropyDecoder {
	eType, err := GetType(name)

	if err != nil || C.knz_supports(0, C.uint32_t(eType)) != 1 {
		return nil
	}

	// The device decoder owns its payload (gpu_entropy.go): everything the test wrote, which ends on a byte.
	payload := make([]byte, 0, 1024)

	for {
		more, err := ibs.HasMoreToRead()

		if err != nil || more == false {
			break
		}

		payload = append(payload, byte(ibs.ReadBits(8)))
	}

	res, err := NewGPUEntropyDecoder(gpuTestOpen(), nil, payload, eType)

	if err != nil {
		panic(err.Error())
	}

	gpuTestObjects++
	return res
}
