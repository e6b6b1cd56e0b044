cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out/r05_final; mkdir -p $O
python bench.py > $O/config_bwt_bench.json 2> $O/bwt.err
python bench.py --config l5 > $O/config_l5_bench.json 2> $O/l5.err
python bench.py --config lz > $O/config_lz_bench.json 2> $O/lz.err
python bench.py --config huffman > $O/config_huffman_bench.json 2> $O/huf.err
python bench.py --config ans0 > $O/config_ans0_bench.json 2> $O/ans0.err
for f in bwt l5 lz huffman ans0; do python -c "
import json; d=json.load(open('$O/config_${f}_bench.json')); c=d['cpu_baseline']; print('$f', d['value'], d['encode_MBps'], d['decode_MBps'], 'cpu', c['encode_MBps'], c['decode_MBps'], c['gpu_encode_over_cpu_encode'], c['gpu_decode_over_cpu_decode'])"; done
