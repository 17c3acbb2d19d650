// Mirror of framework/include/ray.h:9-13.
#pragma once
#include <limits>

#include "cgrt_vec.h"

struct Ray {
    cgrt::vec3 origin{0.0f};
    cgrt::vec3 direction{0.0f, 0.0f, -1.0f};
    float t{std::numeric_limits<float>::max()};
};
static_assert(sizeof(Ray) == 28, "Ray layout (ray.h:9-13)");
