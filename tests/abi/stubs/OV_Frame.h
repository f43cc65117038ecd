/* Stand-in for the camera driver header (reference Drivers/BSP/OV2640/OV_Frame.h:10): the 112x112 RGB565 frame
 * buffer yoloface.c reads.  Defined by tests/abi/yoloface_main.c. */
#ifndef YF_STUB_OV_FRAME_H
#define YF_STUB_OV_FRAME_H
#include <stdint.h>
extern uint8_t RGB_DATA[];
#endif
