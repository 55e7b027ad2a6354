// boost/shared_ptr.hpp -- stand-in, TEST INFRASTRUCTURE ONLY.
#pragma once
#include <memory>
namespace boost { using std::shared_ptr; }
