#!/bin/bash
# all BASELINE configs on one GPU (GPU box): one JSON line per config into gpurun_out/bench_all_TAG.jsonl
# (each line carries its own cpu_baseline: the CPU oracle on the box's host cores, 10-15 s per config; NOCPU=1 skips it)
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
O=gpurun_out/bench_all_$TAG.jsonl
: > $O
for w in robo_unet_640x480_bs32 robo_unet_160x120_bs64 unet_640x480_bs32 labelprop_160x120_b2 labelprop_160x120_b64 robo_unet_320x240_bs32 robo_unet_v2_640x480_bs32; do
  steps=20; if [[ $w == labelprop* ]]; then steps=200; fi
  timeout -k 10 300 python bench.py --workload $w --steps $steps --warmup 5 ${NOCPU:+--no-cpu-baseline} 2> gpurun_out/bench_all_$TAG.err | tail -1 >> $O
done
python - $O <<'PY'
import sys, json
for line in open(sys.argv[1]):
    d = json.loads(line); r = d.get('roofline', {})
    print('%-26s %10.1f img/s %8.3f ms/step  t_roof %7.3f ms  step_frac %.3f  dominant %-26s %s %.1f %s = %.3f of its roof  [%s] cpu %s' % (
        d['config']['workload'], d['value'], d['ms_per_step'], r.get('t_roof_ms', 0), r.get('step_frac', 0), r.get('kernel'), r.get('bound'),
        r.get('achieved', 0), r.get('unit'), r.get('frac', 0), d['config'].get('launch', ''), d.get('cpu_baseline', {}).get('value')))
PY
