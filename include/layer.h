/* Source-compatibility shim: callers of the reference include "layer.h"
 * (src_yolo2/layer.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_LAYER_H
#define SR_YOLO2_SHIM_LAYER_H
#include "sr_yolo2.h"
#endif
