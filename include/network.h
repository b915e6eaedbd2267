/* Source-compatibility shim: callers of the reference include "network.h"
 * (src_yolo2/network.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_NETWORK_H
#define SR_YOLO2_SHIM_NETWORK_H
#include "sr_yolo2.h"
#endif
