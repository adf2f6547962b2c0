// sfmData.hpp — the part of an AliceVision SfMData scene the depth-map stage reads: views, pinhole intrinsics, poses and the
// landmarks with their 2-D observations.  Restates sfmData/{SfMData,View,Landmark,CameraPose}.hpp and the JSON reader
// sfmDataIO/jsonIO.cpp:76-111 (views), :244-449 (intrinsics), :533-566 (landmarks), :707-860 (file) of the reference.
// Alembic (.abc) scenes — what Meshroom's StructureFromMotion node writes — are read by alembic.cpp (no Alembic library needed).
// Rigs are read (a rig camera's pose = its sub-pose composed with the rig's pose).  Not read: ancestors, features/matches folders, constraints.
#pragma once

#include "mvsData.hpp"

#include <cstdint>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace avdm_host {

using IndexT = uint32_t;
static constexpr IndexT UndefinedIndexT = 0xffffffffu;

struct View
{
    IndexT viewId = UndefinedIndexT, poseId = UndefinedIndexT, intrinsicId = UndefinedIndexT;
    // a camera of a rig (sfmData/View.hpp:150-190): its pose is the rig's sub-pose composed with the rig's pose unless it was
    // estimated on its own ("independant")
    IndexT rigId = UndefinedIndexT, subPoseId = UndefinedIndexT;
    bool independantPose = true;
    bool isPartOfRig() const { return rigId != UndefinedIndexT; }
    bool isPoseIndependant() const { return !isPartOfRig() || independantPose; }
    std::string path;
    int width = 0, height = 0;
    std::map<std::string, std::string> metadata;
};

// sfmData/ExposureSetting.hpp: shutter [s], relative aperture, ISO of a view (from its EXIF metadata); -1 = not stated
struct ExposureSetting
{
    double shutter = -1.0, fnumber = -1.0, iso = -1.0;
    bool hasShutter() const;
    bool hasFNumber() const;
    bool isPartiallyDefined() const { return hasShutter() || hasFNumber(); }
    // ExposureSetting::getExposure(referenceISO = 100, referenceFNumber = 1): the exposure time that gives the same light at ISO 100, f/1
    double getExposure() const;
};
// sfmData/ImageInfo.{hpp,cpp}: getCameraExposureSetting() from a view's metadata (keys matched like findMetadataIterator: exact, else
// case-insensitive on the part behind the last '/' or ':'; "1/200"-style fractions)
ExposureSetting cameraExposureSetting(const std::map<std::string, std::string>& metadata);

// camera::Pinhole with camera::IntrinsicScaleOffsetDisto (camera/Pinhole.hpp, IntrinsicScaleOffset.cpp:55-66)
struct Intrinsic
{
    IndexT intrinsicId = UndefinedIndexT;
    std::string type;            // "pinhole", ...
    std::string distortionType;  // "none", "radialk1", "radialk3", ...
    int width = 0, height = 0;
    double sensorWidth = 36.0, sensorHeight = 24.0;
    double scaleX = 1.0, scaleY = 1.0;  // focal length in pixels
    double offsetX = 0.0, offsetY = 0.0;  // principal point offset from the image centre
    std::vector<double> distortionParams;
    bool isPinhole = false;

    Point2d principalPoint() const { return {offsetX + width * 0.5, offsetY + height * 0.5}; }
    Point2d ima2cam(const Point2d& p) const
    {
        const Point2d pp = principalPoint();
        return {(p.x - pp.x) / scaleX, (p.y - pp.y) / scaleY};
    }
    Point2d removeDistortion(const Point2d& p) const;  // camera plane -> camera plane
    Matrix3x3 K() const
    {
        Matrix3x3 k;
        const Point2d pp = principalPoint();
        k(0, 0) = scaleX, k(0, 2) = pp.x, k(1, 1) = scaleY, k(1, 2) = pp.y, k(2, 2) = 1.0;
        return k;
    }
};

// geometry::Pose3: world -> camera rotation and camera centre
struct Pose
{
    Matrix3x3 rotation;
    Point3d center;
};

// sfmData/Rig.hpp: the relative poses of a rig's cameras
struct RigSubPose
{
    bool initialized = false;  // ERigSubPoseStatus != UNINITIALIZED (ESTIMATED and CONSTANT are the same to this stage)
    Pose pose;
};
struct Rig
{
    std::vector<RigSubPose> subPoses;
};

struct Observation
{
    double x = 0.0, y = 0.0;  // full-size image pixels
};
struct Landmark
{
    Point3d X;
    unsigned char rgb[3] = {255, 255, 255};  // image::RGBColor (sfmData/Landmark.hpp): white unless the file says otherwise
    std::map<IndexT, Observation> observations;  // viewId -> observation (ordered like the reference's stl::flat_map)
};

struct SfMData
{
    std::map<IndexT, View> views;  // ordered by viewId like sfmData::Views (HashMap iteration order is unspecified in the
                                   // reference; MultiViewParams only depends on it for the camera index <-> viewId map)
    std::map<IndexT, Intrinsic> intrinsics;
    std::map<IndexT, Pose> poses;
    std::map<IndexT, Landmark> landmarks;

    std::map<IndexT, Rig> rigs;

    const RigSubPose* rigSubPose(const View& v) const
    {
        const auto it = rigs.find(v.rigId);
        return it != rigs.end() && v.subPoseId < it->second.subPoses.size() ? &it->second.subPoses[v.subPoseId] : nullptr;
    }
    // sfmData/SfMData.hpp:288-295 isPoseAndIntrinsicDefined
    bool isPoseAndIntrinsicDefined(const View& v) const
    {
        if(!(v.intrinsicId != UndefinedIndexT && v.poseId != UndefinedIndexT && intrinsics.count(v.intrinsicId) && poses.count(v.poseId)))
            return false;
        if(v.isPoseIndependant())
            return true;
        const RigSubPose* sp = rigSubPose(v);
        return sp != nullptr && sp->initialized;
    }
    // sfmData/SfMData.cpp getPose: the view's own pose, or rig sub-pose * rig pose (geometry::Pose3 composition of [R | -R C] blocks)
    Pose getPose(const View& v) const
    {
        const Pose& base = poses.at(v.poseId);
        if(v.isPoseIndependant())
            return base;
        const RigSubPose* sp = rigSubPose(v);
        if(sp == nullptr)
            throw std::out_of_range("SfMData::getPose: view " + std::to_string(v.viewId) + " names a rig sub-pose that does not exist");
        const Matrix3x3 R = sp->pose.rotation * base.rotation;
        const Point3d tRig = (base.rotation * base.center) * -1.0, tSub = (sp->pose.rotation * sp->pose.center) * -1.0;
        const Point3d t = sp->pose.rotation * tRig + tSub;
        Pose out;
        out.rotation = R;
        out.center = Point3d(-(R(0, 0) * t.x + R(1, 0) * t.y + R(2, 0) * t.z), -(R(0, 1) * t.x + R(1, 1) * t.y + R(2, 1) * t.z),
                             -(R(0, 2) * t.x + R(1, 2) * t.y + R(2, 2) * t.z));
        return out;
    }
    // sfmData/SfMData.hpp:406-426 getMedianCameraExposureSetting().getExposure(): the median over the DISTINCT exposures of the views that state
    // one (-1 when none does: the reference indexes an empty list there)
    double medianCameraExposure() const;
    const Intrinsic& getIntrinsic(const View& v) const { return intrinsics.at(v.intrinsicId); }
};

// sfmDataIO::load for .sfm / .json / .abc; throws std::runtime_error with the reason
void loadSfMData(SfMData& out, const std::string& filename);

// camera::angleBetweenRays (camera/IntrinsicBase.hpp:475-517): degrees between the world rays of two observations
double angleBetweenRays(const Pose& pose1, const Intrinsic& intr1, const Pose& pose2, const Intrinsic& intr2, const Point2d& x1, const Point2d& x2);

} // namespace avdm_host
