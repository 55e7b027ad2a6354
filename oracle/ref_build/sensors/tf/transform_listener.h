// tf/transform_listener.h -- stand-in, TEST INFRASTRUCTURE ONLY: the two types SensorProcessorBase.hpp names.
#pragma once
namespace tf { struct TransformListener {}; struct StampedTransform {}; }
