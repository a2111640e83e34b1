# Builds the gfx950 C-ABI library (libyolo_hip.so). The oracle is Python (the reference is Python): nothing to compile for it.
# hipcc cross-compiles without a GPU present.
HIPCC ?= /opt/rocm/bin/hipcc
ARCH  ?= gfx950
CSRC  := tf2_yolo_amd/csrc
SRCS  := $(CSRC)/runtime.hip $(CSRC)/probe.hip $(CSRC)/conv.hip $(CSRC)/conv_split.hip $(CSRC)/conv_wgrad_split.hip $(CSRC)/conv_planes.hip $(CSRC)/conv_win.hip $(CSRC)/conv_small.hip $(CSRC)/conv_wgrad_planes.hip $(CSRC)/conv_wgrad_win.hip $(CSRC)/stem.hip $(CSRC)/bn_act.hip $(CSRC)/elementwise.hip \
         $(CSRC)/labels.hip $(CSRC)/loss.hip $(CSRC)/decode_nms.hip $(CSRC)/iou.hip $(CSRC)/measure.hip
OBJS  := $(SRCS:.hip=.o)
LIB   := tf2_yolo_amd/libyolo_hip.so
HIPFLAGS := -O3 -std=c++17 -fPIC --offload-arch=$(ARCH) -munsafe-fp-atomics -Wall -Wno-unused-function
ifdef KNOCKOUTS   # diagnostic build: YOLO_PLANES_DBG=<bits> selects compile-time knock-outs of the planes conv loop
HIPFLAGS += -DYOLO_PLANES_KNOCKOUTS
endif

all: $(LIB)

# (every kernel file is rebuilt when ANY shared header changes: act.hpp was missing from this list in round 3 and a change
# to the Mish arithmetic silently never reached bn_act.o)
HDRS  := $(wildcard $(CSRC)/*.hpp) include/yolo_hip.h
$(CSRC)/%.o: $(CSRC)/%.hip $(HDRS)
	$(HIPCC) $(HIPFLAGS) -c $< -o $@

$(LIB): $(OBJS)
	$(HIPCC) -shared -fPIC --offload-arch=$(ARCH) $(OBJS) -ldl -o $@

clean:
	rm -f $(OBJS) $(LIB)

.PHONY: all clean
