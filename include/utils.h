/* Source-compatibility shim: callers of the reference include "utils.h"
 * (src_yolo2/utils.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_UTILS_H
#define SR_YOLO2_SHIM_UTILS_H
#include "sr_yolo2.h"
#endif
