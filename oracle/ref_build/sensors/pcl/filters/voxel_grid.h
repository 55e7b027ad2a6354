// pcl/filters/voxel_grid.h -- stand-in, TEST INFRASTRUCTURE ONLY: named by the point headers, not used by the code under test.
#pragma once
#include <pcl/point_cloud.h>
