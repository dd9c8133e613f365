/* TEST-ONLY stand-in for raylib.h: just the types, constants and prototypes that /root/reference/src/main.c and its
 * headers (inc/body.h, inc/player.h, inc/rand.h) name, so that the reference's main.c can be parsed against
 * the headers under include/ode by tests/test_reference_compiles.py (gcc -fsyntax-only: nothing here is ever linked or run).
 * Written from raylib's public API as main.c uses it; not raylib. */
#pragma once
#include <stdbool.h>
#include <stdarg.h>

#ifndef PI
#define PI 3.14159265358979323846f
#endif
#define DEG2RAD (PI / 180.0f)
#define RAD2DEG (180.0f / PI)

typedef struct Vector2 { float x, y; } Vector2;
typedef struct Vector3 { float x, y, z; } Vector3;
typedef struct Vector4 { float x, y, z, w; } Vector4;
typedef Vector4 Quaternion;
typedef struct Matrix { float m0, m4, m8, m12, m1, m5, m9, m13, m2, m6, m10, m14, m3, m7, m11, m15; } Matrix;
typedef struct Color { unsigned char r, g, b, a; } Color;
typedef struct Rectangle { float x, y, width, height; } Rectangle;
typedef struct Texture { unsigned int id; int width, height, mipmaps, format; } Texture;
typedef Texture Texture2D;
typedef struct RenderTexture { unsigned int id; Texture texture; Texture depth; } RenderTexture;
typedef RenderTexture RenderTexture2D;
typedef struct Shader { unsigned int id; int *locs; } Shader;
typedef struct Camera3D { Vector3 position, target, up; float fovy; int projection; } Camera3D;
typedef Camera3D Camera;
typedef struct Mesh { int vertexCount, triangleCount; float *vertices; } Mesh;
typedef struct MaterialMap { Texture2D texture; Color color; float value; } MaterialMap;
typedef struct Material { Shader shader; MaterialMap *maps; float params[4]; } Material;
typedef struct Model { Matrix transform; int meshCount, materialCount; Mesh *meshes; Material *materials; int *meshMaterial; } Model;

#define WHITE     (Color){ 255, 255, 255, 255 }
#define BLACK     (Color){ 0, 0, 0, 255 }
#define RAYWHITE  (Color){ 245, 245, 245, 255 }
#define GRAY      (Color){ 130, 130, 130, 255 }
#define DARKGRAY  (Color){ 80, 80, 80, 255 }
#define LIGHTGRAY (Color){ 200, 200, 200, 255 }
#define RED       (Color){ 230, 41, 55, 255 }
#define GREEN     (Color){ 0, 228, 48, 255 }
#define BLUE      (Color){ 0, 121, 241, 255 }
#define YELLOW    (Color){ 253, 249, 0, 255 }
#define SKYBLUE   (Color){ 102, 191, 255, 255 }
#define MAGENTA   (Color){ 255, 0, 255, 255 }
#define BLANK     (Color){ 0, 0, 0, 0 }

enum { FLAG_VSYNC_HINT = 0x40, FLAG_MSAA_4X_HINT = 0x20, FLAG_WINDOW_RESIZABLE = 0x4, FLAG_WINDOW_HIGHDPI = 0x2000 };
enum { LOG_ALL = 0, LOG_TRACE, LOG_DEBUG, LOG_INFO, LOG_WARNING, LOG_ERROR, LOG_FATAL, LOG_NONE };
enum { KEY_NULL = 0, KEY_SPACE = 32, KEY_ESCAPE = 256, KEY_ENTER = 257, KEY_TAB = 258, KEY_RIGHT = 262, KEY_LEFT = 263, KEY_DOWN = 264,
       KEY_UP = 265, KEY_LEFT_SHIFT = 340, KEY_LEFT_CONTROL = 341, KEY_RIGHT_SHIFT = 344, KEY_RIGHT_CONTROL = 345,
       KEY_A = 65, KEY_B, KEY_C, KEY_D, KEY_E, KEY_F, KEY_G, KEY_H, KEY_I, KEY_J, KEY_K, KEY_L, KEY_M, KEY_N, KEY_O, KEY_P, KEY_Q, KEY_R,
       KEY_S, KEY_T, KEY_U, KEY_V, KEY_W, KEY_X, KEY_Y, KEY_Z };
enum { MOUSE_BUTTON_LEFT = 0, MOUSE_BUTTON_RIGHT = 1 };
enum { CAMERA_PERSPECTIVE = 0, CAMERA_ORTHOGRAPHIC };
enum { SHADER_LOC_VERTEX_POSITION = 0, SHADER_LOC_MATRIX_MVP = 6, SHADER_LOC_MATRIX_VIEW, SHADER_LOC_MATRIX_PROJECTION, SHADER_LOC_MATRIX_MODEL,
       SHADER_LOC_MATRIX_NORMAL, SHADER_LOC_VECTOR_VIEW, SHADER_LOC_COLOR_DIFFUSE };
enum { SHADER_UNIFORM_FLOAT = 0, SHADER_UNIFORM_VEC2, SHADER_UNIFORM_VEC3, SHADER_UNIFORM_VEC4, SHADER_UNIFORM_INT, SHADER_UNIFORM_IVEC2,
       SHADER_UNIFORM_IVEC3, SHADER_UNIFORM_IVEC4, SHADER_UNIFORM_SAMPLER2D };
enum { TEXTURE_FILTER_POINT = 0, TEXTURE_FILTER_BILINEAR, TEXTURE_FILTER_TRILINEAR };
enum { MATERIAL_MAP_ALBEDO = 0, MATERIAL_MAP_DIFFUSE = 0 };

void InitWindow(int width, int height, const char *title);
void CloseWindow(void);
bool WindowShouldClose(void);
void SetConfigFlags(unsigned int flags);
void SetTargetFPS(int fps);
void SetExitKey(int key);
int GetScreenWidth(void);
int GetScreenHeight(void);
float GetFrameTime(void);
double GetTime(void);
int GetFPS(void);
void TraceLog(int logLevel, const char *text, ...);
const char *TextFormat(const char *text, ...);
int MeasureText(const char *text, int fontSize);
bool IsKeyDown(int key);
bool IsKeyPressed(int key);
bool IsKeyReleased(int key);
bool IsMouseButtonDown(int button);
bool IsMouseButtonPressed(int button);
Vector2 GetMouseDelta(void);
Vector2 GetMousePosition(void);
void DisableCursor(void);
void EnableCursor(void);
void BeginDrawing(void);
void EndDrawing(void);
void BeginMode3D(Camera3D camera);
void EndMode3D(void);
void BeginTextureMode(RenderTexture2D target);
void EndTextureMode(void);
void BeginShaderMode(Shader shader);
void EndShaderMode(void);
void ClearBackground(Color color);
Color GetColor(unsigned int hexValue);
Vector4 ColorNormalize(Color color);
Shader LoadShader(const char *vsFileName, const char *fsFileName);
void UnloadShader(Shader shader);
int GetShaderLocation(Shader shader, const char *uniformName);
void SetShaderValue(Shader shader, int locIndex, const void *value, int uniformType);
void SetShaderValueMatrix(Shader shader, int locIndex, Matrix mat);
void SetTextureFilter(Texture2D texture, int filter);
Mesh GenMeshCube(float width, float height, float length);
Mesh GenMeshSphere(float radius, int rings, int slices);
Mesh GenMeshPlane(float width, float length, int resX, int resZ);
Model LoadModel(const char *fileName);
Model LoadModelFromMesh(Mesh mesh);
void UnloadModel(Model model);
Texture2D LoadTexture(const char *fileName);
void UnloadTexture(Texture2D texture);
void DrawModel(Model model, Vector3 position, float scale, Color tint);
void DrawModelEx(Model model, Vector3 position, Vector3 rotationAxis, float rotationAngle, Vector3 scale, Color tint);
void DrawCube(Vector3 position, float width, float height, float length, Color color);
void DrawCubeWires(Vector3 position, float width, float height, float length, Color color);
void DrawSphere(Vector3 centerPos, float radius, Color color);
void DrawSphereWires(Vector3 centerPos, float radius, int rings, int slices, Color color);
void DrawCylinderEx(Vector3 startPos, Vector3 endPos, float startRadius, float endRadius, int sides, Color color);
void DrawGrid(int slices, float spacing);
void DrawText(const char *text, int posX, int posY, int fontSize, Color color);
void DrawFPS(int posX, int posY);
void DrawRectangle(int posX, int posY, int width, int height, Color color);
void DrawTextureEx(Texture2D texture, Vector2 position, float rotation, float scale, Color tint);
