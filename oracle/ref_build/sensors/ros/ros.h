// ros/ros.h -- stand-in for the sensor-processor build, TEST INFRASTRUCTURE ONLY: a parameter server that knows no parameter (every
// param() call returns its default), a clock, logging macros that do nothing.
#pragma once
#include <string>
#define ROS_DEBUG(...) ((void)0)
#define ROS_INFO(...) ((void)0)
#define ROS_WARN(...) ((void)0)
#define ROS_ERROR(...) ((void)0)
namespace ros {
struct NodeHandle {
    template <class T, class D> bool param(const std::string&, T& value, const D& fallback) const { value = static_cast<T>(fallback); return false; }
};
struct Duration { double s = 0.0; double toSec() const { return s; } };
struct Time { double s = 0.0; static Time now() { return Time(); } Duration operator-(const Time& o) const { Duration d; d.s = s - o.s; return d; } };
}
