/* Source-compatibility shim: callers of the reference include "image.h"
 * (src_yolo2/image.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_IMAGE_H
#define SR_YOLO2_SHIM_IMAGE_H
#include "sr_yolo2.h"
#endif
