/* Source-compatibility shim: callers of the reference include "tree.h"
 * (src_yolo2/tree.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_TREE_H
#define SR_YOLO2_SHIM_TREE_H
#include "sr_yolo2.h"
#endif
