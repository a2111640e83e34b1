for sh in "13 512 1024 3 1 same 32" "26 256 512 3 1 same 32" "52 128 256 3 1 same 32" "104 64 128 3 1 same 32"; do
  for kc in 0 1 2 4 8; do
    echo -n "$sh kc=$kc : "; YOLO_PLANES_KC=$kc timeout -k 10 60 python scripts/one_conv.py $sh 30 fwdp 2>/dev/null | tail -n 1
  done
done
YOLO_PLANES_KC=4 timeout -k 10 200 python -m pytest tests/test_gpu_conv.py -q -k planes 2>&1 | tail -n 1
for kc in 0 4; do echo "traffic kc=$kc"; YOLO_PLANES_KC=$kc bash scripts/pmc_traffic_one.sh 13 512 1024 3 1 same 32 5 fwdp | grep FETCH; done
