/* Source-compatibility shim: callers of the reference include "list.h"
 * (src_yolo2/list.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_LIST_H
#define SR_YOLO2_SHIM_LIST_H
#include "sr_yolo2.h"
#endif
