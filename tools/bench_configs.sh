#!/bin/bash
# all single-GPU BASELINE configs + bookends, one JSON summary line each
run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['config']['workload'][:60], '| ratio', d['config']['compression_ratio'], '| comp', d['per_gpu']['compress_GBps'], 'GB/s', r['launch_ms'],'ms frac', r['frac'], '| decomp', d['per_gpu']['decompress_GBps'], 'GB/s', r['decompress']['launch_ms'], 'ms frac', r['decompress']['frac'], '| exact', d['roundtrip_bit_exact'])"; }
run --shape 512,512,512 --dtype float32
run --shape 8192,8192 --dtype float64
run --shape 16777216 --dtype float32
run --shape 512,512,512 --dtype float64
run --shape 8192,8192 --dtype float32
run --shape 67108864 --dtype float64
run --shape 512,512,512 --dtype float32 --smooth
run --shape 512,512,512 --dtype float32 --data zeros
run --shape 512,512,512 --dtype float32 --data random
run --shape 510,511,509 --dtype float32
run --config 4
# A/B of the 64-bit decoder: decompress_kernel_wide (the lines above ran the library default, 128 work-items per hypercube)
echo "-- f64 decoder with 256 work-items per hypercube (decompress_kernel_wide):"
run --shape 8192,8192 --dtype float64 --f64-work-items 256
run --shape 512,512,512 --dtype float64 --f64-work-items 256
run --shape 67108864 --dtype float64 --f64-work-items 256
run --shape 8192,8192 --dtype float64 --data random --f64-work-items 256
echo "-- (default mapping, random bits 2D f64, for the line above:)"
run --shape 8192,8192 --dtype float64 --data random
for wi in 128 256; do echo -n "cfg 5 slab, $wi work-items: "; python bench.py --config 5 --f64-work-items $wi --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['config']['workload'][:60], '| ratio', d['config']['compression_ratio'], '| decompress-only', d['per_gpu']['decompress_GBps'], 'GB/s', r['launch_ms'], 'ms frac', r['frac'], '| exact', d['roundtrip_bit_exact'])"; done
