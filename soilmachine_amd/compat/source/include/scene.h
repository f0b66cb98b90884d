// compat/source/include/scene.h -- render parameters the host's (stubbed) pipeline lambda names (scene.h:3-25).
#pragma once
namespace scene {
using namespace glm;
vec3 lightcolor = vec3(1);
float lightstrength = 1.0f;
vec3 lightpos = vec3(1.25f, 1.5f, 2.5f);
vec3 skycolor = vec3(0.80, 0.90, 0.89);
bool distancefog = true;
mat4 dp = mat4(1);
mat4 dv = mat4(1);
mat4 bias = mat4(1);
vec3 watercolor = vec3(0.27, 0.5, 0.7);
bool wateroverlay = true;
}  // namespace scene
