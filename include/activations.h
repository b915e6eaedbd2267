/* Source-compatibility shim: callers of the reference include "activations.h"
 * (src_yolo2/activations.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_ACTIVATIONS_H
#define SR_YOLO2_SHIM_ACTIVATIONS_H
#include "sr_yolo2.h"
#endif
