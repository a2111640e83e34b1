#!/bin/bash
# Data-parallel readiness on ONE GPU: usage (GPU box, repo root): bash scripts/dp_readiness.sh <out.json>
#  (1) the tape-mode step plain vs with RCCL in a forced world of one rank (YOLO_DP_FORCE=1): every collective of the N > 1 job
#  (2) per-bucket ready / done times from the start of backward
#  (3) two gloo ranks sharing the GPU at bs 16 each: per-rank step time of the rehearsal
OUT=${1:-gpurun_out/dp_readiness.json}; T=$(mktemp -d)
A="--steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-extra-blocks"
for i in 1 2; do
  python bench.py $A 2>/dev/null | tail -1 > $T/plain_$i.json
  YOLO_DP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=2957$i RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py $A 2>/dev/null | tail -1 > $T/forced_$i.json
done
YOLO_BENCH_SINGLE_DEVICE=1 YOLO_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29579 \
  bench.py --gpus 2 --batch 16 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer --no-extra-blocks 2>/dev/null | grep '^{' | tail -1 > $T/gloo2.json
python - $T $OUT <<'PY'
import json, sys
T, out = sys.argv[1], sys.argv[2]
ld = lambda n: json.load(open(f"{T}/{n}.json"))
plain = [ld(f"plain_{i}")["ms_per_step"] for i in (1, 2)]
forced = [ld(f"forced_{i}") for i in (1, 2)]
g = ld("gloo2")
res = {"what": "YOLOv3-416 bs 32 training step, launch-tape mode, same box, alternating runs (plain, forced, plain, forced)",
       "plain_ms_per_step": plain, "rccl_world1_forced_ms_per_step": [f["ms_per_step"] for f in forced],
       "overhead_pct": round((sum(f["ms_per_step"] for f in forced) / sum(plain) - 1) * 100, 2),
       "replicas_in_sync": [f["config"]["replicas_in_sync"] for f in forced],
       "bucket_trace_rccl_world1": forced[-1].get("dp_trace"),
       "gloo_2_ranks_one_gpu_bs16_each": {"ms_per_step": g["ms_per_step"], "images_per_s": g["value"],
                                          "replicas_in_sync": g["config"]["replicas_in_sync"], "dp_trace": g.get("dp_trace")}}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:3000])
PY
