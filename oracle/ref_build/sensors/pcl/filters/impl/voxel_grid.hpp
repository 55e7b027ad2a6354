// pcl/filters/impl/voxel_grid.hpp -- stand-in, TEST INFRASTRUCTURE ONLY: nothing of it is used by the code under test.
#pragma once
