/* Source-compatibility shim: callers of the reference include "box.h"
 * (src_yolo2/box.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_BOX_H
#define SR_YOLO2_SHIM_BOX_H
#include "sr_yolo2.h"
#endif
