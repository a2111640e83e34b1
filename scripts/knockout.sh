# usage: bash scripts/knockout.sh <one_conv args...>: the planes conv with parts of its main loop knocked out
# (needs the diagnostic build: make KNOCKOUTS=1). bits: 1 no DMA, 2 no fragment reads, 4 no barrier, 8 no MFMA, 16 no stores
for Z in 0 1; do for D in 0 1 2 3 7 8 11 15 16 23 31; do
  echo "zeros=$Z dbg=$D: $(YOLO_ONE_CONV_ZEROS=$Z YOLO_PLANES_DBG=$D python3 scripts/one_conv.py "$@" 2>/dev/null | tail -n 1)"
done; done
