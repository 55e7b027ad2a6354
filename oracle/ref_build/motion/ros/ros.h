// ros/ros.h -- stand-in for the motion build (oracle/ref_build), TEST INFRASTRUCTURE ONLY: the two ROS types RobotMotionMapUpdater touches.
#pragma once
#include <string>
namespace ros {
struct Time {
    double t;
    Time() : t(0) {}
    explicit Time(double v) : t(v) {}
    static Time now() { return Time(-1.0); }
    bool operator==(const Time& o) const { return t == o.t; }
    double toSec() const { return t; }
};
extern double gemref_param_covariance_scale;
struct NodeHandle {
    template <class T> bool param(const std::string&, T& out, const T&) const { out = static_cast<T>(gemref_param_covariance_scale); return true; }
};
}
