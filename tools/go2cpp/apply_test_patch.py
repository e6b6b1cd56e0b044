#!/usr/bin/env python3
"""The hooks that re-point the reference's OWN unit tests at the device objects, applied to COPIES of three of its test files (test infrastructure:
oracle/_ref_gpu_tests). Insertions only, nothing is removed or rewritten:
  entropy/Entropy_test.go       getEncoder / getDecoder: first statement hands out gpuTestEncoder / gpuTestDecoder (go/testhooks) when it has one
  transform/Transforms_test.go  getTransform: first statement hands out gpuTestTransform when it has one
  io/CompressedStream_test.go   compress: the Writer and the Reader are asked for the device path right after they are built, and give it back after Close
usage: apply_test_patch.py entropy|transform|io SRC DST"""
import os
import sys

which, src_path, dst_path = sys.argv[1:4]
src = open(src_path).read()


def insert_after(text, anchor, addition):
    if text.count(anchor) != 1:
        sys.exit(f"apply_test_patch: {anchor.strip()!r} not found exactly once in {src_path}")
    return text.replace(anchor, anchor + addition)


if which == "entropy":
    src = insert_after(src, "func getEncoder(name string, obs kanzi.OutputBitStream) kanzi.EntropyEncoder {\n",
                       "\tif enc := gpuTestEncoder(name, obs); enc != nil {\n\t\treturn enc\n\t}\n\n")
    src = insert_after(src, "func getDecoder(name string, ibs kanzi.InputBitStream) kanzi.EntropyDecoder {\n",
                       "\tif dec := gpuTestDecoder(name, ibs); dec != nil {\n\t\treturn dec\n\t}\n\n")
elif which == "transform":
    src = insert_after(src, "func getTransform(name string) (kanzi.ByteTransform, error) {\n",
                       "\tif tf := gpuTestTransform(name); tf != nil {\n\t\treturn tf, nil\n\t}\n\n")
elif which == "io":
    src = insert_after(src, "\t\t// Compress block\n", "\t\tgpuTestEnableWriter(w)\n")
    src = insert_after(src, "\t\t// Decompress block\n", "\t\tgpuTestEnableReader(r)\n")
    src = insert_after(src, "\t\t// Close Writer\n\t\terr = w.Close()\n", "\t\tw.DisableGPU()\n")
    src = insert_after(src, "\t\t// Close Reader\n\t\terr = r.Close()\n", "\t\tr.DisableGPU()\n")
else:
    sys.exit("apply_test_patch: entropy | transform | io")
os.makedirs(os.path.dirname(os.path.abspath(dst_path)), exist_ok=True)
open(dst_path, "w").write(src)
