#pragma once
#include <memory>
namespace pcl { struct PointXYZRGB {}; template <class T> struct PointCloud { typedef std::shared_ptr<PointCloud<T>> Ptr; }; }
