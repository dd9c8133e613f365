/* TEST-ONLY stand-in for rlgl.h (see raylib.h in this directory): the calls main.c's shadow-map code names. */
#pragma once
#include "raylib.h"
enum { RL_ATTACHMENT_COLOR_CHANNEL0 = 0, RL_ATTACHMENT_DEPTH = 100, RL_ATTACHMENT_TEXTURE2D = 100, RL_ATTACHMENT_RENDERBUFFER = 200 };
enum { RL_SHADER_UNIFORM_FLOAT = 0, RL_SHADER_UNIFORM_INT = 4 };
#define TRACELOG(level, ...) TraceLog(level, __VA_ARGS__)
void rlPushMatrix(void);
void rlPopMatrix(void);
void rlMultMatrixf(const float *matf);
Matrix rlGetMatrixModelview(void);
Matrix rlGetMatrixProjection(void);
unsigned int rlLoadFramebuffer(void);
unsigned int rlLoadTextureDepth(int width, int height, bool useRenderBuffer);
void rlFramebufferAttach(unsigned int fboId, unsigned int texId, int attachType, int texType, int mipLevel);
bool rlFramebufferComplete(unsigned int id);
void rlUnloadFramebuffer(unsigned int id);
void rlEnableFramebuffer(unsigned int id);
void rlDisableFramebuffer(void);
void rlEnableShader(unsigned int id);
void rlActiveTextureSlot(int slot);
void rlEnableTexture(unsigned int id);
void rlSetUniform(int locIndex, const void *value, int uniformType, int count);
