#!/bin/bash
# compact register / scratch / LDS table of every kernel of one .hip file: scripts/kernel_resources.sh tf2_yolo_amd/csrc/conv_win.hip [filter]
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/kr_$$.o 2>&1 \
  | grep -E "Function Name|VGPRs:|ScratchSize|Occupancy \[|LDS Size" | sed -e 's/.*remark: [^ ]* *//' -e 's/ \[-Rpass.*//' | paste - - - - - \
  | sed -e 's/Function Name: //' | { if [ -n "$2" ]; then grep "$2"; else cat; fi; } | while read -r l; do n=$(echo "$l" | cut -f1 | c++filt | cut -c1-90); echo "$n | $(echo "$l" | cut -f2-)"; done
rm -f /tmp/kr_$$.o
