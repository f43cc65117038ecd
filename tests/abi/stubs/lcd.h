/* Stand-in for the board's LCD driver header, declaring only what stm32/X-CUBE-AI/App/yoloface.c uses (own text;
 * the reference's Drivers/BSP/LCD/lcd.h:153,222 declare the same two names).  Test scaffolding for the boundary
 * binary oracle/_ref/abi_yoloface_caller: it lets the reference's UNMODIFIED yoloface.c compile off the MCU. */
#ifndef YF_STUB_LCD_H
#define YF_STUB_LCD_H
#include <stdint.h>
#define RED 0xF800
void LCD_DrawRectangle(uint16_t x1, uint16_t y1, uint16_t x2, uint16_t y2, uint16_t Color);
#endif
