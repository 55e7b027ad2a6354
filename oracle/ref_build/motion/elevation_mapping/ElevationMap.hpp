// elevation_mapping/ElevationMap.hpp -- stand-in that SHADOWS the reference's header in the motion build (oracle/ref_build), TEST
// INFRASTRUCTURE ONLY: RobotMotionMapUpdater::update reads the map's size and pose and nothing else (RMU.cpp:51, 63).
#pragma once
#include <kindr/Core>
#include <ros/ros.h>
namespace grid_map {
struct Size { int v[2]; int operator()(int i) const { return v[i]; } };
struct GridMap { Size s; const Size& getSize() const { return s; } };
}
namespace elevation_mapping {
class ElevationMap {
 public:
    grid_map::GridMap raw; kindr::HomogeneousTransformationPosition3RotationQuaternionD pose;
    grid_map::GridMap& getRawGridMap() { return raw; }
    const kindr::HomogeneousTransformationPosition3RotationQuaternionD& getPose() const { return pose; }
};
}
