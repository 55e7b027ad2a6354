// pcl/filters/impl/passthrough.hpp -- stand-in, TEST INFRASTRUCTURE ONLY: nothing of it is used by the code under test.
#pragma once
