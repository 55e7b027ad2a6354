// pcl/filters/passthrough.h -- stand-in, TEST INFRASTRUCTURE ONLY: the z pass-through the structured-light processor's cleanPointCloud
// configures (limits inclusive, non-finite points dropped, as in PCL).
#pragma once
#include <cmath>
#include <string>
#include <pcl/point_cloud.h>
namespace pcl {
template <class P> struct PassThrough {
    typename PointCloud<P>::ConstPtr in; std::string field; double lo = -1e300, hi = 1e300;
    void setInputCloud(const typename PointCloud<P>::ConstPtr& c) { in = c; }
    void setFilterFieldName(const std::string& f) { field = f; }
    void setFilterLimits(double a, double b) { lo = a; hi = b; }
    void filter(PointCloud<P>& out)
    {
        PointCloud<P> r;
        for (const P& p : in->points) {
            if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
            const float v = field == "x" ? p.x : field == "y" ? p.y : p.z;
            if (v < lo || v > hi) continue;
            r.points.push_back(p);
        }
        r.width = (std::uint32_t)r.points.size(); r.height = 1;
        out.swap(r);
    }
};
}
