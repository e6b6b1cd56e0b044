// gpu_hooks_transform_test.go — goes into github.com/flanglet/kanzi-go/v2/transform next to Transforms_test.go (test build only).
//
// Transforms_test.go builds every transform it tests through getTransform (Transforms_test.go:44-101). With the one-line
// hook at its top (tools/go2cpp/apply_test_patch.py) it hands out the device-backed object of gpu_transform.go for the
// transforms the library implements, and the reference's own tests (TestLZ, TestLZX, TestLZP, TestZRLT, TestSRT, TestRank,
// TestMTFT, TestTextCodec, TestUTFCodec, TestLZCodecSpecifics, ...) run against the device unchanged.
package transform

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

// GpuTestObjects is the number of device-backed transform objects the tests were handed so far.
func GpuTestObjects() int {
	return gpuTestObjects
}

func gpuTestTransform(name string) kanzi.ByteTransform {
	id := uint64(0)

	switch name {
	case "LZ":
		id = LZ_TYPE
	case "LZX":
		id = LZX_TYPE
	case "LZP":
		id = LZP_TYPE
	case "ZRLT":
		id = ZRLT_TYPE
	case "SRT":
		id = SRT_TYPE
	case "RANK":
		id = RANK_TYPE
	case "MTFT":
		id = MTFT_TYPE
	case "TEXT":
		id = DICT_TYPE
	case "UTF":
		id = UTF_TYPE
	default:
		return nil
	}

	if gpuTestHandle == nil {
		var cfg C.knz_cfg
		cfg.block_size = C.uint32_t(4 << 20)
		cfg.bs_version = 6
		cfg.device = -1

		if rc := C.knz_open(&cfg, &gpuTestHandle); rc != 0 {
			panic("Cannot open the device: " + C.GoString(C.knz_last_error(nil)))
		}
	}

	res, err := NewGPUTransform(gpuTestHandle, id)

	if err != nil {
		panic(err.Error())
	}

	gpuTestObjects++
	return res
}
