// pcl/point_cloud.h -- stand-in, TEST INFRASTRUCTURE ONLY: the container the sensor processors read.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>
namespace pcl {
template <class P> struct PointCloud {
    typedef std::shared_ptr<PointCloud<P>> Ptr;
    typedef std::shared_ptr<const PointCloud<P>> ConstPtr;
    std::vector<P> points;
    std::uint32_t width = 0, height = 0;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    void swap(PointCloud& o) { points.swap(o.points); std::swap(width, o.width); std::swap(height, o.height); std::swap(is_dense, o.is_dense); }
};
}
