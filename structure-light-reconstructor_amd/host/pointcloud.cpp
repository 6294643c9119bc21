// pointcloud.cpp -- PointCloudImage (Duke/pointcloudimage.cpp:3-164) and the pattern encoders (Duke/graycodes.cpp:22-128,
// Duke/multifrequency.cpp:14-33) of the host mirror.  Host-side bookkeeping only.
#include "duke.hpp"

#include <math.h>

#include <fstream>

namespace duke {

PointCloudImage::PointCloudImage(int imageW, int imageH, bool colorFlag) : w(imageW), h(imageH)
{
    points.assign((size_t)w * h * 3, 0.0f);
    if (colorFlag) color.assign((size_t)w * h * 3, 0);
    numOfPointsForPixel.assign((size_t)w * h, 0);
}
bool PointCloudImage::setPoint(int i_w, int j_h, Point3f p)
{
    if (i_w >= w || j_h >= h) return false;
    float *d = &points[((size_t)j_h * w + i_w) * 3];
    d[0] = p.x; d[1] = p.y; d[2] = p.z;
    numOfPointsForPixel[(size_t)j_h * w + i_w] = 1;
    return true;
}
bool PointCloudImage::setPoint(int i_w, int j_h, Point3f p, const int cg[3])
{
    if (i_w >= w || j_h >= h) return false;
    setPoint(i_w, j_h, p);
    if (!color.empty())
        for (int k = 0; k < 3; k++) color[((size_t)j_h * w + i_w) * 3 + k] = (uint8_t)(cg[k] < 0 ? 0 : cg[k] > 255 ? 255 : cg[k]);
    return true;
}
bool PointCloudImage::getPoint(int i_w, int j_h, Point3f &out) const
{
    if (i_w >= w || j_h >= h) return false;
    const uint8_t num = numOfPointsForPixel[(size_t)j_h * w + i_w];
    if (num == 0) return false;
    const float *s = &points[((size_t)j_h * w + i_w) * 3];
    const double dn = (double)(float)num;                    // Vec3d / float (pointcloudimage.cpp:62)
    out.x = (float)((double)s[0] / dn); out.y = (float)((double)s[1] / dn); out.z = (float)((double)s[2] / dn);
    return true;
}
bool PointCloudImage::getPoint(int i_w, int j_h, Point3f &out, int colorOut[3]) const
{
    if (!getPoint(i_w, j_h, out)) return false;
    const uint8_t num = numOfPointsForPixel[(size_t)j_h * w + i_w];
    if (!color.empty())
        for (int k = 0; k < 3; k++) colorOut[k] = (int)lrint((double)color[((size_t)j_h * w + i_w) * 3 + k] / (double)(float)num);
    else { colorOut[0] = colorOut[1] = colorOut[2] = 100; }  // pointcloudimage.cpp:49 "(100,100,100)" comma expression -> x=100
    return true;
}
bool PointCloudImage::addPoint(int i_w, int j_h, Point3f p)
{
    if (i_w >= w || j_h >= h) return false;
    const size_t o = (size_t)j_h * w + i_w;
    const uint8_t num = numOfPointsForPixel[o];
    if (num == 0) return setPoint(i_w, j_h, p);
    points[o * 3] = p.x + points[o * 3]; points[o * 3 + 1] = p.y + points[o * 3 + 1]; points[o * 3 + 2] = p.z + points[o * 3 + 2];
    numOfPointsForPixel[o] = (uint8_t)(num + 1);
    return true;
}
bool PointCloudImage::addPoint(int i_w, int j_h, Point3f p, const int cg[3])
{
    if (i_w >= w || j_h >= h) return false;
    const size_t o = (size_t)j_h * w + i_w;
    if (numOfPointsForPixel[o] == 0) return setPoint(i_w, j_h, p, cg);
    addPoint(i_w, j_h, p);
    if (color.empty()) return false;
    for (int k = 0; k < 3; k++) { int v = cg[k] + color[o * 3 + k]; color[o * 3 + k] = (uint8_t)(v > 255 ? 255 : v); }
    return true;
}
void PointCloudImage::exportXYZ(const char *path, bool exportOffPixels, bool colorFlag) const
{
    std::ofstream out(path);
    for (int i = 0; i < w; i++)
        for (int j = 0; j < h; j++) {
            const uint8_t num = numOfPointsForPixel[(size_t)j * w + i];
            if (!exportOffPixels && num == 0) continue;
            Point3f p; int c[3] = {0, 0, 0};
            getPoint(i, j, p, c);
            if (exportOffPixels && num == 0) { p = Point3f(); c[0] = c[1] = c[2] = 0; }
            out << p.x << " " << p.y << " " << p.z;
            if (colorFlag && !color.empty()) out << " " << c[2] << " " << c[1] << " " << c[0] << "\n";
            else out << "\n";
        }
}


// ------------------------------------------------------------------------------------------------------------
// encoders
// ------------------------------------------------------------------------------------------------------------
GrayCodes::GrayCodes(int scanW, int scanH, bool useepi) : useEpi(useepi), height(scanH), width(scanW) { calNumOfImgs(); }
void GrayCodes::calNumOfImgs()
{
    numOfColImgs = (int)ceil(log((double)width) / log(2.0));
    numOfRowImgs = (int)ceil(log((double)height) / log(2.0));
    numOfImgs = useEpi ? 2 + 2 * numOfColImgs : 2 + 2 * numOfColImgs + 2 * numOfRowImgs;
}
void GrayCodes::generateGrays()
{
    grayCodes.assign(numOfImgs, Image8());
    for (auto &g : grayCodes) { g.w = width; g.h = height; g.d.assign((size_t)width * height, 0); }
    std::fill(grayCodes[0].d.begin(), grayCodes[0].d.end(), 255);
    for (int j = 0; j < width; j++) {
        int num = j, prevRem = j % 2;
        for (int k = 0; k < numOfColImgs; k++) {
            num /= 2;
            const int rem = num % 2;
            const uint8_t a = (rem != prevRem) ? 255 : 0, b = a ? 0 : 255;
            Image8 &pa = grayCodes[2 * numOfColImgs - 2 * k], &pb = grayCodes[2 * numOfColImgs - 2 * k + 1];
            for (int i = 0; i < height; i++) { pa.d[(size_t)i * width + j] = a; pb.d[(size_t)i * width + j] = b; }
            prevRem = rem;
        }
    }
    if (!useEpi)
        for (int i = 0; i < height; i++) {
            int num = i, prevRem = i % 2;
            for (int k = 0; k < numOfRowImgs; k++) {
                num /= 2;
                const int rem = num % 2;
                const uint8_t a = (rem != prevRem) ? 255 : 0, b = a ? 0 : 255;
                Image8 &pa = grayCodes[2 * numOfRowImgs - 2 * k + 2 * numOfColImgs];
                Image8 &pb = grayCodes[2 * numOfRowImgs - 2 * k + 2 * numOfColImgs + 1];
                for (int j = 0; j < width; j++) { pa.d[(size_t)i * width + j] = a; pb.d[(size_t)i * width + j] = b; }
                prevRem = rem;
            }
        }
}
int GrayCodes::grayToDec(const std::vector<bool> &gray)
{
    int dec = 0;
    bool tmp = gray[0];
    const int n = (int)gray.size();
    if (tmp) dec += 1 << (n - 1);
    for (int i = 1; i < n; i++) {
        tmp = (tmp != gray[i]);
        if (tmp) dec += 1 << (n - i - 1);
    }
    return dec;
}

MultiFrequency::MultiFrequency(int projwidth, int projheight) : projW(projwidth), projH(projheight) {}
void MultiFrequency::generateMutiFreq()
{
    static const int frequency[3] = {70, 64, 59};
    const double PI = 3.1416;                               // multifrequency.h:5
    for (int i = 0; i < 14; i++) { MultiFreqImages[i].w = projW; MultiFreqImages[i].h = projH; MultiFreqImages[i].d.assign((size_t)projW * projH, i == 0 ? 255 : 0); }
    for (int f = 0; f < 3; f++)
        for (int phi = 0; phi < 4; phi++) {
            Image8 &t = MultiFreqImages[4 * f + phi + 2];
            for (int x = 0; x < projW; x++) {
                const float v = 135 + 79 * cosf((float)(PI * 2 * (double)x * (double)frequency[f] / (double)projW + PI * (double)phi / 2));
                const uint8_t g = (uint8_t)v;
                for (int y = 0; y < projH; y++) t.d[(size_t)y * projW + x] = g;
            }
        }
}


}  // namespace duke
