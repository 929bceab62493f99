/* lkingest.h -- C ABI of the video ingest: frames of an ISO-BMFF (.mp4) file with one H.264 Constrained-Baseline
 * track, for hosts that have no cv::VideoCapture (this image has no video decoder of any kind).  Replaces, for the file the reference
 * ships (pacman.mp4), the frame source of the reference's loop:
 *
 *     cv::VideoCapture cap(...); cap >> frame;                    ParallelVideoPyr.cpp:769-776
 *
 * Host-only code (optical-flow_amd/ingest/h264_ingest.cpp, built by g++ into optical-flow_amd/liblkingest.so): nothing here touches the
 * GPU or the timed path; the frames enter the flow library as 8-bit planes (lkflow_stream_push_u8, lkflow_compute_u8).  The decoder is a
 * second, scalar implementation of optical-flow_amd/h264_ingest.py (same clauses of ITU-T H.264); the two agree byte for byte on the
 * reference's file (tests/test_h264_ingest.py).  Decoder parity against a reference decoder is UNPINNED (none exists here): DESIGN.md 2.
 *
 * Not handled (LKINGEST_E_UNSUPPORTED): CABAC, B slices, interlace, slice groups, 8x8 transform, weighted prediction, long-term
 * references, more than one slice per picture.
 */
#ifndef LKINGEST_H
#define LKINGEST_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LKINGEST_OK 0
#define LKINGEST_E_ARG (-1)
#define LKINGEST_E_IO (-2)           /* file cannot be read / is not an mp4 with an avc1 track */
#define LKINGEST_E_UNSUPPORTED (-3)  /* a coding tool outside the list above */
#define LKINGEST_E_STREAM (-4)       /* the bit stream does not parse (every slice must end exactly at its stop bit) */

typedef struct lkingest lkingest;

/* cv::VideoCapture::open(path): reads the file, its sample tables and parameter sets. */
int lkingest_open(lkingest **out, const char *path);
/* CV_CAP_PROP_FRAME_HEIGHT / _WIDTH / _FRAME_COUNT */
int lkingest_info(const lkingest *v, int *rows, int *cols, int *n_frames);
/* Decodes the next picture (decoding order = output order: Baseline has no B pictures).  Any destination may be NULL.
 * y: rows x cols luma, cb / cr: rows/2 x cols/2.  Returns 1 if a picture was produced, 0 at the end of the stream, < 0 on error. */
int lkingest_read_yuv(lkingest *v, unsigned char *y, size_t y_step, unsigned char *cb, unsigned char *cr, size_t c_step);
/* cap >> frame: the next picture as interleaved BGR (CV_8UC3), BT.601 limited range, chroma replicated 2 x 2
 * (== h264_ingest.yuv420_to_bgr).  Same return values. */
int lkingest_read_bgr(lkingest *v, unsigned char *bgr, size_t step);
void lkingest_close(lkingest *v);
const char *lkingest_last_error(const lkingest *v);

#ifdef __cplusplus
}
#endif
#endif
