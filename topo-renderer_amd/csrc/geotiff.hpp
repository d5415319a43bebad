// geotiff.hpp -- container side of the GeoTIFF decode that precedes the render path (SURVEY.md 8f rank 2;
// reference: fetch_terrain, topo-renderer/src/control/background_runner.rs:99-136, which hands the bytes to the `tiff`
// crate 0.11.2 -- Decoder::new / find_tag / read_image_to_buffer / dimensions -- and the three geo tags to
// CoordinateTransform::from_geo_tag_data).  The crate is not part of the reference tree; what is restated here is the
// published format: TIFF 6.0 (classic, both byte orders), strips or tiles, one 32-bit IEEE float sample per pixel,
// compression none / Deflate (8 and the legacy 32946) / LZW (5) / PackBits (32773), predictor none / horizontal (2,
// 32-bit words) / floating point (3, Adobe Photoshop TIFF Technical Note 3).
//
// Host: parse the first IFD, undo the byte-stream compression of every strip/tile (zlib for Deflate).  Device
// (topo_kernels.hip: k_tiff_rows): undo the predictor, fix the byte order, place the segments into the w x h raster --
// the samples never exist as floats on the host.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

namespace topo {

struct TiffSegment {            // one strip or tile
    uint64_t offset = 0, bytes = 0;     // in the file
    uint32_t x0 = 0, y0 = 0;            // position in the image
    uint32_t w = 0, h = 0;              // stored size (tiles are padded to the full tile size; strips have the image width)
};

struct TiffInfo {
    bool big_endian = false;
    uint32_t width = 0, height = 0;
    uint32_t compression = 1, predictor = 1;
    std::vector<TiffSegment> segments;
    // geo tags (absent: empty vector)
    std::vector<double> pixel_scale, tie_points, model_transformation;
    bool has_pixel_scale = false, has_tie_points = false, has_model_transformation = false;
};

// 0 on success, else a topo_status (TOPO_ERR_INVALID: malformed, TOPO_ERR_UNSUPPORTED: a feature outside the list above).
int tiff_parse(const uint8_t* data, size_t n, TiffInfo& info, std::string& err);

// CoordinateTransform::from_geo_tag_data on the file's geo tags (terrain_renderer.cpp).
int geotiff_transform(const TiffInfo& info, float raster_point[2], float model_point[2], float pixel_scale[2]);

// The decompressed (still predicted) bytes of one segment: exactly seg.w * seg.h * 4 bytes.
int tiff_segment_bytes(const uint8_t* data, size_t n, const TiffInfo& info, const TiffSegment& seg, uint8_t* out, std::string& err);

}  // namespace topo
