#!/usr/bin/env python3
"""The patch of INTEGRATION.md section 1, applied to a COPY of the reference's v2/io/CompressedStream.go (test infrastructure: oracle/_ref_gpu).
Two insertions, nothing else: the first statement of Writer.processBlock and of Reader.processBlock hands the batch to the GPU scheduler when the
stream owns one (go/gpu_stream.go). usage: apply_integration_patch.py SRC DST"""
import os
import sys

src = open(sys.argv[1]).read()
for head, hook in (("func (this *Writer) processBlock() error {\n", "gpuBatchOfWriter"), ("func (this *Reader) processBlock() (int64, error) {\n", "gpuBatchOfReader")):
    if src.count(head) != 1:
        sys.exit(f"apply_integration_patch: {head.strip()!r} not found exactly once")
    src = src.replace(head, head + f"\tif gb := {hook}(this); gb != nil {{\n\t\treturn this.processBlockGPU(gb)\n\t}}\n\n")
os.makedirs(os.path.dirname(os.path.abspath(sys.argv[2])), exist_ok=True)
open(sys.argv[2], "w").write(src)
