// pcl/pcl_macros.h -- stand-in for the sensor-processor build, TEST INFRASTRUCTURE ONLY (PCL is not installed here): the field
// macros of the reference's point structs, laid out as PCL lays them out; registration / instantiation macros expand to nothing.
#pragma once
#include <cstdint>
#include <Eigen/Core>
#define PCL_ADD_POINT4D union { float data[4]; struct { float x; float y; float z; }; }
#define PCL_ADD_RGB union { union { struct { std::uint8_t b; std::uint8_t g; std::uint8_t r; std::uint8_t a; }; float rgb; }; std::uint32_t rgba; }
#define POINT_CLOUD_REGISTER_POINT_STRUCT(name, ...) static_assert(sizeof(name) > 0, "")
#define PCL_INSTANTIATE(what, type) static_assert(sizeof(type) > 0, "")
