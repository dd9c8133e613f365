/* TEST-ONLY stand-in for raygui.h (see raylib.h in this directory): the controls and style properties main.c's menu names.
 * (The reference vendors the real single-header raygui under inc/; this directory comes first on the include path so that
 * the 6 000-line implementation, which needs all of raylib, is not what gets parsed.) */
#pragma once
#include "raylib.h"
enum { DEFAULT = 0, LABEL, BUTTON, TOGGLE, SLIDER, PROGRESSBAR, CHECKBOX, COMBOBOX, DROPDOWNBOX, TEXTBOX };
enum { BORDER_COLOR_NORMAL = 0, BASE_COLOR_NORMAL, TEXT_COLOR_NORMAL, BORDER_COLOR_FOCUSED, BASE_COLOR_FOCUSED, TEXT_COLOR_FOCUSED,
       BORDER_WIDTH = 12, TEXT_PADDING, TEXT_ALIGNMENT };
enum { TEXT_SIZE = 16, TEXT_SPACING, LINE_COLOR, BACKGROUND_COLOR, TEXT_LINE_SPACING };
void GuiSetStyle(int control, int property, int value);
int GuiGetStyle(int control, int property);
void GuiLoadStyleJungle(void);
int GuiLabel(Rectangle bounds, const char *text);
int GuiButton(Rectangle bounds, const char *text);
int GuiTextBox(Rectangle bounds, char *text, int textSize, bool editMode);
