/* Stand-in for Drivers/BSP/.../Picture.h: yoloface.c includes it but uses nothing from it. */
#ifndef YF_STUB_PICTURE_H
#define YF_STUB_PICTURE_H
#endif
