/* Source-compatibility shim: callers of the reference include "parser.h"
 * (src_yolo2/parser.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_PARSER_H
#define SR_YOLO2_SHIM_PARSER_H
#include "sr_yolo2.h"
#endif
