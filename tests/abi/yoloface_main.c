/* Boundary check, row a15 of SURVEY.md 8(a): the reference's UNMODIFIED application file
 * stm32/X-CUBE-AI/App/yoloface.c (aiInit, aiRun, resize_rgb565_uint8_112_to_56_direct, prepare_yolo_data, post_process)
 * compiled where it lies, together with the reference's network_data.c, against the reference's own AI headers and
 * linked to libyf_network.so in place of network.c + the ST runtime.  This file plays stm32/User/main.c:24-56: the
 * same call order and the same two printf lines around each frame, with the camera replaced by a file of frames.
 *
 *   abi_yoloface_caller <rgb565_frames.bin> <n> <in_data_out.bin> <out_data_out.bin>     (UART text on stdout)
 * This is a test of the boundary (link + call sequence + data formats), not an oracle pin: the network behind
 * ai_network_run is this repository's GPU engine. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "yoloface.h"
#include "lcd.h"
#include "network.h"

uint8_t RGB_DATA[112 * 112 * 2];      /* the camera frame buffer (reference Drivers/BSP/OV2640/OV_Frame.c:29) */
uint32_t frame = 0;                   /* stm32/User/main.c:20-21 */
uint8_t face_num = 0;
extern ai_i8 in_data[AI_NETWORK_IN_1_SIZE];     /* non-static in yoloface.c:10-15 */
extern ai_i8 out_data[AI_NETWORK_OUT_1_SIZE];

static long rectangles = 0;
void LCD_DrawRectangle(uint16_t x1, uint16_t y1, uint16_t x2, uint16_t y2, uint16_t Color) {
  (void)x1; (void)y1; (void)x2; (void)y2; (void)Color;
  ++rectangles;
}

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s rgb565_frames.bin n in_data_out.bin out_data_out.bin\n", argv[0]); return 2; }
  const int n = atoi(argv[2]);
  FILE* fi = fopen(argv[1], "rb");
  FILE* f_in = fopen(argv[3], "wb");
  FILE* f_out = fopen(argv[4], "wb");
  if (!fi || !f_in || !f_out) { fprintf(stderr, "cannot open files\n"); return 2; }
  if (aiInit() != 0) return 4;                                       /* main.c:38 */
  for (int k = 0; k < n; ++k) {
    if (fread(RGB_DATA, sizeof RGB_DATA, 1, fi) != 1) { fprintf(stderr, "short frame file\n"); return 2; }   /* GetImage() */
    face_num = 0;                                                    /* main.c:44-53 */
    frame++;
    printf("=== Frame %d ===\r\n----------------------------------------\r\n", (int)frame);
    resize_rgb565_uint8_112_to_56_direct();
    prepare_yolo_data();
    if (aiRun() != 0) return 5;
    post_process();
    printf("----------------------------------------\r\n[INFO] Total faces detected: %d\r\n", face_num);
    fwrite(in_data, 1, AI_NETWORK_IN_1_SIZE, f_in);
    fwrite(out_data, 1, AI_NETWORK_OUT_1_SIZE, f_out);
  }
  fclose(fi); fclose(f_in); fclose(f_out);
  fflush(stdout);
  fprintf(stderr, "%d frames, %ld rectangles drawn\n", n, rectangles);
  return 0;
}
