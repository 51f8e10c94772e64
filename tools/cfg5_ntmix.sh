#!/bin/bash
# cfg 5 with the masked rows written back through L2 (nt_mix bit 0x100) instead of streamed: do the fix-up repairs then hit lines still cached?  (ablation build: HRX_NT_MIX is read there only)
P='import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d["roofline"]
print("ms/step %.4f frac %.3f verified %s" % (d["ms_per_step"], r["frac"], (d.get("verified") or {}).get("bit_exact")))'
B="python3 bench.py --config dfa256 --len 4095 --rows 4096 --warmup 3 --no-cpu-baseline --no-pmc --no-spread --batch 131072 --steps 10 --allow-debug-flags"
export HRX_LIB_PATH=halo2_regex_amd/csrc/libhrx_ablation.so
for p in 0 20 200; do for m in 0 0x100; do echo -n "pairs $p nt_mix $m: "; HRX_NT_MIX=$m timeout 300 $B --substr-pairs $p 2>/dev/null | python3 -c "$P"; done; done
