// refgen: writes .knz streams with the REFERENCE implementation (flanglet/kanzi-go v2) for the cases listed in a manifest.
//
// This is the one piece of the parity evidence that cannot be produced in the build image (no Go toolchain there): run
// tools/make_ref_vectors.sh on any machine that has Go and a kanzi-go checkout, commit tests/golden/ref_streams/, and
// tests/test_ref_streams.py then requires oracle-encode == file, device-encode == file and device-decode(file) == input.
// It only uses the reference's public API (v2/io.NewWriterWithCtx, the same ctx keys the CLI fills in, app/BlockCompressor.go).
package main

import (
	"bytes"
	"encoding/json"
	"fmt"
	"io"
	"os"
	"path/filepath"

	kio "github.com/flanglet/kanzi-go/v2/io"
)

type refCase struct {
	Name      string `json:"name"`
	Input     string `json:"input"`
	Transform string `json:"transform"`
	Entropy   string `json:"entropy"`
	BlockSize uint   `json:"block_size"`
	Checksum  uint   `json:"checksum"`
	Skip      bool   `json:"skip_blocks"`
}

type manifest struct {
	Cases []refCase `json:"cases"`
}

func encode(c refCase, inDir, outDir string) error {
	data, err := os.ReadFile(filepath.Join(inDir, c.Input))
	if err != nil {
		return err
	}
	out, err := os.Create(filepath.Join(outDir, c.Name+".knz"))
	if err != nil {
		return err
	}
	ctx := make(map[string]any)
	ctx["entropy"] = c.Entropy
	ctx["transform"] = c.Transform
	ctx["blockSize"] = c.BlockSize
	ctx["jobs"] = uint(1)
	ctx["checksum"] = c.Checksum
	ctx["fileSize"] = int64(len(data))
	ctx["headerless"] = false
	if c.Skip {
		ctx["skipBlocks"] = true
	}
	w, err := kio.NewWriterWithCtx(out, ctx)
	if err != nil {
		return err
	}
	if _, err = w.Write(data); err != nil {
		return err
	}
	if err = w.Close(); err != nil {
		return err
	}
	// the reference must read its own stream back
	in, err := os.Open(filepath.Join(outDir, c.Name+".knz"))
	if err != nil {
		return err
	}
	r, err := kio.NewReader(in, 1)
	if err != nil {
		return err
	}
	back, err := io.ReadAll(r)
	if err != nil {
		return err
	}
	r.Close()
	if !bytes.Equal(back, data) {
		return fmt.Errorf("%s: the reference does not round-trip its own stream", c.Name)
	}
	return nil
}

func main() {
	if len(os.Args) != 4 {
		fmt.Fprintln(os.Stderr, "usage: refgen <manifest.json> <input dir> <output dir>")
		os.Exit(2)
	}
	raw, err := os.ReadFile(os.Args[1])
	if err != nil {
		fmt.Fprintln(os.Stderr, err)
		os.Exit(1)
	}
	var m manifest
	if err = json.Unmarshal(raw, &m); err != nil {
		fmt.Fprintln(os.Stderr, err)
		os.Exit(1)
	}
	failed := 0
	for _, c := range m.Cases {
		if err = encode(c, os.Args[2], os.Args[3]); err != nil {
			fmt.Fprintf(os.Stderr, "FAILED %s: %v\n", c.Name, err)
			failed++
			continue
		}
		fmt.Printf("ok %s\n", c.Name)
	}
	if failed != 0 {
		os.Exit(1)
	}
}
