/* Source-compatibility shim: callers of the reference include "test_detector.h"
 * (src_yolo2/test_detector.h); every declaration now lives in sr_yolo2.h. */
#ifndef SR_YOLO2_SHIM_TEST_DETECTOR_H
#define SR_YOLO2_SHIM_TEST_DETECTOR_H
#include "sr_yolo2.h"
#endif
