# round 6, the last tree: the five bench lines with bench.py's new defaults (10 timed steps behind 3 untimed ones; with 3 behind 1 the default line moved by 4 % from run to run)
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r06_last_lines; mkdir -p $O
python bench.py > $O/config_bwt_bench.json 2> $O/bwt.err
python bench.py --config l5 > $O/config_l5_bench.json 2> $O/l5.err
python bench.py --config lz > $O/config_lz_bench.json 2> $O/lz.err
python bench.py --config huffman > $O/config_huffman_bench.json 2> $O/huf.err
python bench.py --config ans0 > $O/config_ans0_bench.json 2> $O/ans0.err
for f in bwt l5 lz huffman ans0; do cut -c1-190 $O/config_${f}_bench.json; done
