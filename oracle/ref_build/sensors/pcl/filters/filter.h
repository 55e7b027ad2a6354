// pcl/filters/filter.h -- stand-in, TEST INFRASTRUCTURE ONLY: removeNaNFromPointCloud as PCL defines it (finite x, y, z kept, their
// indices reported).  cleanPointCloud is not what the test is after; the stereo model reads the indices it leaves.
#pragma once
#include <cmath>
#include <vector>
#include <pcl/point_cloud.h>
namespace pcl {
template <class P> void removeNaNFromPointCloud(const PointCloud<P>& in, PointCloud<P>& out, std::vector<int>& index)
{
    PointCloud<P> r; index.clear();
    for (size_t i = 0; i < in.points.size(); ++i) {
        const P& p = in.points[i];
        if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
        r.points.push_back(p); index.push_back((int)i);
    }
    r.width = (std::uint32_t)r.points.size(); r.height = 1; r.is_dense = true;
    out.swap(r);
}
}
