/* A pure C host of libyolo_hip.so (no Python, no torch): includes include/yolo_hip.h as C99, checks the ABI version and
 * that the library reports argument errors through its status codes / yolo_last_error(), and -- when a GPU is present and
 * "run" is given -- enqueues one real call (yolo_fill on a hipMalloc'ed buffer is left to hosts that link the HIP runtime;
 * this file deliberately needs nothing but the C-ABI).
 *   gcc -std=c99 -Iinclude examples/c_host/abi_check.c -o abi_check -Ltf2_yolo_amd -lyolo_hip -Wl,-rpath,$PWD/tf2_yolo_amd
 * The reference (samson6460/tf2_YOLO) is Python and has no FFI: this is what a compiled host of the
 * path binds instead of the ctypes table of tf2_yolo_amd/_lib.py. */
#include <stdio.h>
#include <string.h>

#include "yolo_hip.h"

int main(void) {
  const int abi = yolo_abi_version();
  printf("abi %d\n", abi);
  if (abi < 5) return 1;
  /* a call with null pointers must come back as YOLO_ERR_INVALID_ARG with a message, without touching the device */
  const int rc = yolo_fill(NULL, 16, 0.0f, NULL);
  printf("yolo_fill(NULL) -> %d (%s)\n", rc, yolo_last_error());
  if (rc != YOLO_ERR_INVALID_ARG || strlen(yolo_last_error()) == 0) return 2;
  if (yolo_set_option(12345, 0) != YOLO_ERR_INVALID_ARG) return 3;
  if (yolo_set_option(-1, 0) != YOLO_OK) return 4;
  printf("planes bytes of a 416x416x32 activation at bs 32: %lld\n", (long long)yolo_planes_bytes(32LL * 416 * 416, 32));
  printf("device available: %d\n", yolo_device_available());
  return 0;
}
