// gpu_stream.go — goes into github.com/flanglet/kanzi-go/v2/io next to CompressedStream.go and gpu_batch.go.
//
// The two batch hooks themselves: Writer.processBlockGPU / Reader.processBlockGPU do what Writer.processBlock
// (CompressedStream.go:621-710) and Reader.processBlock (:1614-1744) do, with the goroutine fan-out replaced by ONE call
// into the GPU batch scheduler (gpu_batch.go). The reference's structs stay as they are: which Writer / Reader owns a
// batch scheduler is kept in a side table. The only lines that change in CompressedStream.go are the first statement of
// each processBlock (INTEGRATION.md, "the patch"):
//
//	func (this *Writer) processBlock() error {
//		if gb := gpuBatchOfWriter(this); gb != nil { return this.processBlockGPU(gb) }
//	func (this *Reader) processBlock() (int64, error) {
//		if gb := gpuBatchOfReader(this); gb != nil { return this.processBlockGPU(gb) }
//
// No Go toolchain exists in the image this library is built in. This file, gpu_batch.go and the patched
// CompressedStream.go ARE executed there all the same: tools/go2cpp translates them to C++ together with the rest of the
// reference (`make -C oracle _ref_gpu`), the result is linked against libknz_gpu.so and the GPU suite requires that the
// reference's own Writer / Reader, driving the device through this shim, write and read the same streams as without it
// (tests/test_go_shim_gpu.py).
package io

import (
	"fmt"
	"sync"
	"sync/atomic"
	"time"

	kanzi "github.com/flanglet/kanzi-go/v2"
)

var gpuLock sync.Mutex
var gpuWriters = make(map[*Writer]*gpuBatch)
var gpuReaders = make(map[*Reader]*gpuBatch)

func gpuBatchOfWriter(w *Writer) *gpuBatch {
	gpuLock.Lock()
	gb := gpuWriters[w]
	gpuLock.Unlock()
	return gb
}

func gpuBatchOfReader(r *Reader) *gpuBatch {
	gpuLock.Lock()
	gb := gpuReaders[r]
	gpuLock.Unlock()
	return gb
}

// EnableGPU re-points the block batches of this Writer at the GPU batch scheduler. To be called right after
// NewWriterWithCtx. Returns an error (and leaves the goroutine path in place) when the stream's transform sequence or
// entropy codec has no device implementation or no GPU can be opened: the library has no CPU fallback.
func (this *Writer) EnableGPU() error {
	return this.enableGPU(nil)
}

// EnableGPUDevices is EnableGPUDepth over several GPUs: `devices` are HIP ordinals (GPUDeviceCount tells how many there
// are), every batch of `depth` blocks is cut into len(devices) contiguous balanced ranges and each device encodes its own
// range (upload, kernels, download) at the same time as the others, inside the one knz_encode_blocks call of the batch.
// Blocks are independent (Definitions.go:73-77): the stream written is the same for every list of devices. An ordinal
// may be named several times: its lanes share the device and overlap each other's copies and kernels. depth <= 0 keeps
// the Writer's `jobs`. To be called before the first Write.
func (this *Writer) EnableGPUDevices(devices []int, depth int) error {
	if len(devices) == 0 || len(devices) > _MAX_CONCURRENCY {
		return &IOError{msg: "The number of GPU lanes must be in [1..64]", code: kanzi.ERR_INVALID_PARAM}
	}

	if depth <= 0 {
		depth = this.jobs
	}

	return this.enableGPUDepth(depth, devices)
}

func (this *Writer) enableGPU(devices []int) error {
	if gpuSupports(this.transformType, this.entropyType) == false {
		return &IOError{msg: "No device implementation for this transform / entropy combination", code: kanzi.ERR_INVALID_CODEC}
	}

	checksum := 0

	if this.hasher32 != nil {
		checksum = 32
	} else if this.hasher64 != nil {
		checksum = 64
	}

	skipBlocks := false

	if v, hasKey := this.ctx["skipBlocks"]; hasKey {
		skipBlocks, _ = v.(bool)
	}

	gb, err := newGPUBatch(this.transformType, this.entropyType, this.blockSize, checksum, skipBlocks, devices)

	if err != nil {
		return err
	}

	gpuLock.Lock()
	gpuWriters[this] = gb
	gpuLock.Unlock()
	return nil
}

// EnableGPUDepth is EnableGPU with a batch depth of its own: `depth` blocks (1..1024) are buffered and handed to the
// device per batch instead of `jobs` (which the reference caps at 64, _MAX_CONCURRENCY: it is a number of goroutines
// there; here it is only how many blocks are in flight on the device, and the chains of the BWT pipelines want hundreds:
// DESIGN.md section 4.9). To be called before the first Write. The stream written is the same for
// every depth. Host memory: depth x 2 block buffers, allocated as they fill; device workspace grows with the batch and
// the library takes a batch in halves when the device cannot hold it.
func (this *Writer) EnableGPUDepth(depth int) error {
	return this.enableGPUDepth(depth, nil)
}

func (this *Writer) enableGPUDepth(depth int, devices []int) error {
	if depth < 1 || depth > 1024 {
		return &IOError{msg: "The batch depth must be in [1..1024]", code: kanzi.ERR_INVALID_PARAM}
	}

	if this.available != 0 || atomic.LoadInt32(&this.blockID) != 0 {
		return &IOError{msg: "The batch depth must be set before the first Write", code: kanzi.ERR_INVALID_PARAM}
	}

	if err := this.enableGPU(devices); err != nil {
		return err
	}

	if depth != this.jobs {
		first := this.buffers[0]
		this.jobs = depth
		this.buffers = make([]blockBuffer, 2*depth)
		this.buffers[0] = first

		for i := 1; i < 2*depth; i++ {
			this.buffers[i] = blockBuffer{Buf: make([]byte, 0)}
		}
	}

	return nil
}

// DisableGPU gives the batch scheduler back (to be called after Close).
func (this *Writer) DisableGPU() {
	gpuLock.Lock()
	gb := gpuWriters[this]
	delete(gpuWriters, this)
	gpuLock.Unlock()

	if gb != nil {
		gb.close()
	}
}

// processBlockGPU = Writer.processBlock with the per-block goroutines replaced by one device batch.
// A panic of the shared bitstream (a failing io.Writer under it) comes back as IOError{ERR_PROCESS_BLOCK}, as from an
// encodingTask (:735-743), and any error behind the header cancels the stream's block counter (:745-747).
func (this *Writer) processBlockGPU(gb *gpuBatch) (err error) {
	if err = this.writeHeader(); err != nil {
		return err
	}

	if this.available == 0 {
		return nil
	}

	defer func() {
		if r := recover(); r != nil {
			switch v := r.(type) {
			case error:
				err = &IOError{msg: v.Error(), code: kanzi.ERR_PROCESS_BLOCK}
			default:
				err = &IOError{msg: fmt.Sprint(v), code: kanzi.ERR_PROCESS_BLOCK}
			}
		}

		if err != nil {
			atomic.StoreInt32(&this.blockID, _CANCEL_TASKS_ID)
		}
	}()

	data := make([][]byte, 0, this.jobs)
	out := make([][]byte, 0, this.jobs)
	lengths := make([]int, 0, this.jobs)
	// an encodingTask's output buffer: room for a block the transforms and the entropy coder expanded (:857-864)
	bufSize := max(this.blockSize+(this.blockSize>>3), 256*1024)

	for taskID := 0; taskID < this.jobs; taskID++ {
		dataLength := this.available

		if dataLength > this.blockSize {
			dataLength = this.blockSize
		}

		if dataLength == 0 {
			break
		}

		this.available -= dataLength

		if len(this.buffers[this.jobs+taskID].Buf) < bufSize {
			this.buffers[this.jobs+taskID].Buf = make([]byte, bufSize)
		}

		data = append(data, this.buffers[taskID].Buf)
		out = append(out, this.buffers[this.jobs+taskID].Buf)
		lengths = append(lengths, dataLength)
	}

	res, encErr := gb.encodeBlocks(data, lengths, out)

	if encErr != nil {
		return encErr
	}

	firstID := int(atomic.LoadInt32(&this.blockID))
	hashType := kanzi.EVT_HASH_NONE

	if this.hasher32 != nil {
		hashType = kanzi.EVT_HASH_32BITS
	} else if this.hasher64 != nil {
		hashType = kanzi.EVT_HASH_64BITS
	}

	// ordered emission into the shared bitstream, as the tasks do it one after the other (:934-976); in front of each
	// block the events its task would have sent (:766-771, :850-855, :889-894, :916-931), in the order a one-job Writer
	// sends them, with the same ids, sizes and hashes
	for i := range res {
		if len(this.listeners) > 0 {
			id := firstID + i + 1
			notifyListeners(this.listeners, kanzi.NewEvent(kanzi.EVT_BEFORE_TRANSFORM, id, int64(lengths[i]), res[i].checksum, hashType, time.Now()))
			notifyListeners(this.listeners, kanzi.NewEvent(kanzi.EVT_AFTER_TRANSFORM, id, int64(res[i].postLen), res[i].checksum, hashType, time.Now()))
			notifyListeners(this.listeners, kanzi.NewEvent(kanzi.EVT_BEFORE_ENTROPY, id, int64(res[i].postLen), res[i].checksum, hashType, time.Now()))
			notifyListeners(this.listeners, kanzi.NewEvent(kanzi.EVT_AFTER_ENTROPY, id, int64((res[i].written+7)>>3), res[i].checksum, hashType, time.Now()))

			if v, hasKey := this.ctx["verbosity"]; hasKey {
				if v.(uint) > 4 {
					msg := fmt.Sprintf("{ \"type\":\"%s\", \"id\":%d, \"offset\":%d, \"skipFlags\":%.8b }", "BLOCK_INFO", id, this.obs.Written(), res[i].skipFlags)
					notifyListeners(this.listeners, kanzi.NewEventFromString(kanzi.EVT_BLOCK_INFO, id, msg, time.Now()))
				}
			}
		}

		emitBlocks(this.obs, out[i:i+1], res[i:i+1])
	}

	atomic.AddInt32(&this.blockID, int32(len(lengths)))
	return nil
}

// EnableGPU re-points the block batches of this Reader at the GPU batch scheduler. The stream header is read first:
// the codecs, the block size and the checksum size come from it.
func (this *Reader) EnableGPU() error {
	return this.enableGPU(nil)
}

// EnableGPUDevices is EnableGPUDepth over several GPUs (see Writer.EnableGPUDevices): up to `depth` blocks are read from
// the stream per batch and decoded on len(devices) devices side by side. depth <= 0 keeps the Reader's `jobs`.
func (this *Reader) EnableGPUDevices(devices []int, depth int) error {
	if len(devices) == 0 || len(devices) > _MAX_CONCURRENCY {
		return &IOError{msg: "The number of GPU lanes must be in [1..64]", code: kanzi.ERR_INVALID_PARAM}
	}

	if depth <= 0 {
		depth = this.jobs
	}

	return this.enableGPUDepth(depth, devices)
}

func (this *Reader) enableGPU(devices []int) error {
	if err := this.readHeader(); err != nil {
		return err
	}

	if gpuSupports(this.transformType, this.entropyType) == false {
		return &IOError{msg: "No device implementation for this transform / entropy combination", code: kanzi.ERR_INVALID_CODEC}
	}

	checksum := 0

	if this.hasher32 != nil {
		checksum = 32
	} else if this.hasher64 != nil {
		checksum = 64
	}

	gb, err := newGPUBatch(this.transformType, this.entropyType, this.blockSize, checksum, false, devices)

	if err != nil {
		return err
	}

	gpuLock.Lock()
	gpuReaders[this] = gb
	gpuLock.Unlock()
	return nil
}

// EnableGPUDepth is EnableGPU with a batch depth of its own (see Writer.EnableGPUDepth): up to `depth` blocks are read
// from the stream and decoded per device batch. To be called before the first Read.
func (this *Reader) EnableGPUDepth(depth int) error {
	return this.enableGPUDepth(depth, nil)
}

func (this *Reader) enableGPUDepth(depth int, devices []int) error {
	if depth < 1 || depth > 1024 {
		return &IOError{msg: "The batch depth must be in [1..1024]", code: kanzi.ERR_INVALID_PARAM}
	}

	if this.available != 0 || this.consumed != 0 || atomic.LoadInt32(&this.blockID) != 0 {
		return &IOError{msg: "The batch depth must be set before the first Read", code: kanzi.ERR_INVALID_PARAM}
	}

	if err := this.enableGPU(devices); err != nil {
		return err
	}

	if depth != this.jobs {
		this.jobs = depth
		this.buffers = make([]blockBuffer, 2*depth)

		for i := range this.buffers {
			this.buffers[i] = blockBuffer{Buf: make([]byte, 0)}
		}
	}

	return nil
}

// DisableGPU gives the batch scheduler back (to be called after Close).
func (this *Reader) DisableGPU() {
	gpuLock.Lock()
	gb := gpuReaders[this]
	delete(gpuReaders, this)
	gpuLock.Unlock()

	if gb != nil {
		gb.close()
	}
}

// processBlockGPU = Reader.processBlock: the payloads are read from the shared bitstream one after the other as the
// decoding tasks do (:1816-1852), then ONE device batch replaces the concurrent part of the tasks (:1875-2011).
// Block n of the blocks that are decoded is left in this.buffers[n].Buf, where Reader.Read picks it up (:1713). Blocks
// outside ctx["from"] / ctx["to"] are read and dropped as the tasks do (:1854-1867); a batch that holds nothing else is
// followed by the next one (:1736-1739). A panic of the shared bitstream (a truncated stream: "No more data to read in
// the bitstream") comes back as IOError{ERR_PROCESS_BLOCK}, as from a decodingTask (:1778-1786), and every error cancels
// the stream's block counter (:1789-1791).
func (this *Reader) processBlockGPU(gb *gpuBatch) (decoded int64, err error) {
	if atomic.LoadInt32(&this.blockID) == _CANCEL_TASKS_ID {
		return 0, nil
	}

	defer func() {
		if r := recover(); r != nil {
			if e, ok := r.(error); ok {
				err = &IOError{msg: e.Error(), code: kanzi.ERR_PROCESS_BLOCK}
			} else {
				err = &IOError{msg: "Unknown error", code: kanzi.ERR_PROCESS_BLOCK}
			}
		}

		if err != nil {
			atomic.StoreInt32(&this.blockID, _CANCEL_TASKS_ID)
		}
	}()

	bufSize := this.blockSize + _EXTRA_BUFFER_SIZE

	if bufSize < this.blockSize+(this.blockSize>>4) {
		bufSize = this.blockSize + (this.blockSize >> 4)
	}

	from, to := 0, 0x7FFFFFFF

	if v, hasKey := this.ctx["from"]; hasKey {
		from = v.(int)
	}

	if v, hasKey := this.ctx["to"]; hasKey {
		to = v.(int)
	}

	for {
		payload := make([][]byte, 0, this.jobs)
		out := make([][]byte, 0, this.jobs)
		ids := make([]int, 0, this.jobs)
		firstID := int(atomic.LoadInt32(&this.blockID))
		offsets := make([]uint64, 0, this.jobs)
		endOfStream := false
		read1 := 0

		for taskID := 0; taskID < this.jobs; taskID++ {
			blockOffset := this.ibs.Read()
			lr := uint(this.ibs.ReadBits(5)) + 3
			read := this.ibs.ReadBits(lr)

			if read == 0 {
				// end of stream: nothing is decoded from here on (what the first task that reads an empty block does, :1781-1793)
				atomic.StoreInt32(&this.blockID, _CANCEL_TASKS_ID)
				endOfStream = true
				break
			}

			if read > uint64(1)<<34 {
				return 0, &IOError{msg: "Invalid block size", code: kanzi.ERR_BLOCK_SIZE}
			}

			r := int((read + 7) >> 3)

			if len(this.buffers[this.jobs+taskID].Buf) < r {
				this.buffers[this.jobs+taskID].Buf = make([]byte, r)
			}

			data := this.buffers[this.jobs+taskID].Buf

			// Read data from shared bitstream
			for n := uint(0); read > 0; {
				chkSize := uint(1 << 30)

				if read < 1<<30 {
					chkSize = uint(read)
				}

				this.ibs.ReadArray(data[n:], chkSize)
				n += ((chkSize + 7) >> 3)
				read -= uint64(chkSize)
			}

			atomic.AddInt32(&this.blockID, 1)
			read1++
			id := firstID + taskID + 1

			if id < from || id >= to {
				continue // Check if the block must be skipped (:1854-1867)
			}

			k := len(payload)

			if len(this.buffers[k].Buf) < bufSize {
				this.buffers[k].Buf = make([]byte, bufSize)
			}

			payload = append(payload, data[0:r])
			out = append(out, this.buffers[k].Buf)
			ids = append(ids, id)
			offsets = append(offsets, blockOffset)
		}

		if len(payload) == 0 {
			if endOfStream {
				this.notifyEndGPU(firstID + read1 + 1)
				break
			}

			if read1 != 0 {
				continue // Unless all blocks were skipped, exit the loop (usual case) (:1736-1739)
			}

			break
		}

		sizes, decErr := gb.decodeBlocks(payload, out)

		if decErr != nil {
			return 0, decErr
		}

		for i := range sizes {
			if sizes[i] > this.blockSize {
				return decoded, &IOError{msg: "Block incorrectly decompressed", code: kanzi.ERR_PROCESS_BLOCK}
			}

			decoded += int64(sizes[i])

			if len(this.listeners) > 0 {
				this.notifyBlockGPU(ids[i], payload[i], sizes[i], offsets[i])
			}
		}

		if endOfStream {
			this.notifyEndGPU(firstID + read1 + 1)
		}

		break
	}

	this.consumed = 0
	return decoded, nil
}

// notifyBlockGPU sends the events a decodingTask sends for one block (CompressedStream.go:1905-1931, :1960-1971 and the
// in-order EVT_AFTER_TRANSFORM of Reader.processBlock, :1722-1733), with the same ids, sizes and hashes: the fields they
// carry are the first whole bytes of the block's payload (mode, skip flags, length, checksum: :1875-1916).
func (this *Reader) notifyBlockGPU(id int, data []byte, decoded int, blockOffset uint64) {
	pos := 0
	mode := data[pos]
	pos++
	skipFlags := byte(0)

	if mode&_COPY_BLOCK_MASK == 0 {
		if mode&_TRANSFORMS_MASK != 0 {
			skipFlags = data[pos]
			pos++
		} else {
			skipFlags = (mode << 4) | 0x0F
		}
	}

	dataSize := 1 + int((mode>>5)&0x03)
	preTransformLength := uint64(0)

	for k := 0; k < dataSize; k++ {
		preTransformLength = (preTransformLength << 8) | uint64(data[pos])
		pos++
	}

	hashType := kanzi.EVT_HASH_NONE
	hashBytes := 0

	if this.hasher32 != nil {
		hashType = kanzi.EVT_HASH_32BITS
		hashBytes = 4
	} else if this.hasher64 != nil {
		hashType = kanzi.EVT_HASH_64BITS
		hashBytes = 8
	}

	checksum1 := uint64(0)

	for k := 0; k < hashBytes; k++ {
		checksum1 = (checksum1 << 8) | uint64(data[pos])
		pos++
	}

	if v, hasKey := this.ctx["verbosity"]; hasKey {
		if v.(uint) > 4 {
			msg := fmt.Sprintf("{ \"type\":\"%s\", \"id\":%d, \"offset\":%d, \"skipFlags\":%.8b }", "BLOCK_INFO", id, blockOffset, skipFlags)
			notifyListeners(this.listeners, kanzi.NewEventFromString(kanzi.EVT_BLOCK_INFO, id, msg, time.Now()))
		}
	}

	notifyListeners(this.listeners, kanzi.NewEvent(kanzi.EVT_BEFORE_ENTROPY, id, int64(len(data)), checksum1, hashType, time.Now()))
	notifyListeners(this.listeners, kanzi.NewEvent(kanzi.EVT_AFTER_ENTROPY, id, int64(preTransformLength), checksum1, hashType, time.Now()))
	notifyListeners(this.listeners, kanzi.NewEvent(kanzi.EVT_BEFORE_TRANSFORM, id, int64(preTransformLength), checksum1, hashType, time.Now()))
	notifyListeners(this.listeners, kanzi.NewEvent(kanzi.EVT_AFTER_TRANSFORM, id, int64(decoded), checksum1, hashType, time.Now()))
}

// notifyEndGPU: the task that reads the end-of-stream marker decodes nothing, and Reader.processBlock still reports its
// (empty) result to the listeners (:1700-1733): one EVT_AFTER_TRANSFORM of size 0 with the id behind the last block.
func (this *Reader) notifyEndGPU(id int) {
	if len(this.listeners) == 0 {
		return
	}

	hashType := kanzi.EVT_HASH_NONE

	if this.hasher32 != nil {
		hashType = kanzi.EVT_HASH_32BITS
	} else if this.hasher64 != nil {
		hashType = kanzi.EVT_HASH_64BITS
	}

	notifyListeners(this.listeners, kanzi.NewEvent(kanzi.EVT_AFTER_TRANSFORM, id, 0, 0, hashType, time.Now()))
}
