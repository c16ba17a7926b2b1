// sq_host_types.h — the objects behind the opaque handles of include/squigly_host.h, shared by the
// host build (sq_host.cpp) and the GPU build (sq_bih_device.hip).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/squigly_host.h"

struct sq_mesh {
    std::vector<sq_tri> tris;          // loader order (src/Obj.hs:73-86)
    std::vector<sq_material> mats;
    std::string show_first_object, show_materials;   // `print (head objs)` / `print mats` of --debug (src/Obj.hs:55-57); empty unless loaded from text
};

struct sq_bih {
    sq_bounds root;
    std::vector<sq_node> nodes;        // pre-order (BIH.flatten, src/BIH.hs:50-52)
    std::vector<sq_tri> tris;          // leaf order
    std::vector<sq_material> mats;
    int32_t height = 0, leaves = 0, longest = 0;
};
