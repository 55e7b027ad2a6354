// pcl/kdtree/impl/kdtree_flann.hpp -- stand-in, TEST INFRASTRUCTURE ONLY: nothing of it is used by the code under test.
#pragma once
