// gpu_hooks_io_test.go — goes into github.com/flanglet/kanzi-go/v2/io next to CompressedStream_test.go (test build only).
//
// TestCompressedStream (CompressedStream_test.go:29-96) round-trips random blocks through a Writer and a Reader over a
// temporary file, with random jobs and block sizes and a 32-bit checksum, in its compress helper (:98-186). With the hooks
// of tools/go2cpp/apply_test_patch.py the helper asks both for the device path (gpu_stream.go EnableGPU) right after it
// builds them; streams whose codecs have no device implementation (NONE+ROLZ) stay on the goroutine path.
package io

var gpuTestStreams int

// GpuTestStreams is the number of Writers and Readers of the tests that ran their block batches on the device so far.
func GpuTestStreams() int {
	return gpuTestStreams
}

func gpuTestEnableWriter(w *Writer) {
	if err := w.EnableGPU(); err == nil {
		gpuTestStreams++
	}
}

func gpuTestEnableReader(r *Reader) {
	if err := r.EnableGPU(); err == nil {
		gpuTestStreams++
	}
}
