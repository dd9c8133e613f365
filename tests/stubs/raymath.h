/* TEST-ONLY stand-in for raymath.h (see raylib.h in this directory): the vector / matrix helpers main.c names. */
#pragma once
#include "raylib.h"
typedef struct float16 { float v[16]; } float16;
Vector3 Vector3Zero(void);
Vector3 Vector3One(void);
Vector3 Vector3Add(Vector3 v1, Vector3 v2);
Vector3 Vector3Subtract(Vector3 v1, Vector3 v2);
Vector3 Vector3Scale(Vector3 v, float scalar);
Vector3 Vector3Normalize(Vector3 v);
Vector3 Vector3Negate(Vector3 v);
Vector3 Vector3CrossProduct(Vector3 v1, Vector3 v2);
float Vector3Length(const Vector3 v);
float Vector3DotProduct(Vector3 v1, Vector3 v2);
Vector3 Vector3Lerp(Vector3 v1, Vector3 v2, float amount);
Matrix MatrixIdentity(void);
Matrix MatrixMultiply(Matrix left, Matrix right);
Matrix MatrixTranslate(float x, float y, float z);
Matrix MatrixScale(float x, float y, float z);
Matrix MatrixRotateXYZ(Vector3 angle);
float16 MatrixToFloatV(Matrix mat);
#define MatrixToFloat(mat) (MatrixToFloatV(mat).v)
float Clamp(float value, float min, float max);
float Lerp(float start, float end, float amount);
