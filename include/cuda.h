/* Source-compatibility shim: callers of the reference include "cuda.h"
 * (src_yolo2/cuda.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_CUDA_H
#define SR_YOLO2_SHIM_CUDA_H
#include "sr_yolo2.h"
#endif
