/* Source-compatibility shim: callers of the reference include "option_list.h"
 * (src_yolo2/option_list.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_OPTION_LIST_H
#define SR_YOLO2_SHIM_OPTION_LIST_H
#include "sr_yolo2.h"
#endif
