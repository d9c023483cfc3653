/* The four libsamplerate entry points the reference encoder calls, as a pass-through: the fixture signal is generated
 * at the DCS sample rate (31 250 Hz), so the conversion ratio is 1 and there is nothing to resample.  (The vendored
 * libsamplerate cannot be built: its high_qual_coeffs.h is listed in .MISSING_LARGE_BLOBS.)  Fixture generator only. */
#include <stdlib.h>
#include <string.h>
#include "samplerate.h"
struct SRC_STATE_tag { int unused; };
SRC_STATE *src_new(int converter_type, int channels, int *error) { (void)converter_type; (void)channels; if (error) *error = 0; return (SRC_STATE *)calloc(1, sizeof(struct SRC_STATE_tag)); }
SRC_STATE *src_delete(SRC_STATE *state) { free(state); return NULL; }
int src_set_ratio(SRC_STATE *state, double new_ratio) { (void)state; return new_ratio == 1.0 ? 0 : 1; }
int src_process(SRC_STATE *state, SRC_DATA *data)
{
    (void)state;
    long n = data->input_frames < data->output_frames ? data->input_frames : data->output_frames;
    memcpy(data->data_out, data->data_in, (size_t)n * sizeof(float));
    data->input_frames_used = n;
    data->output_frames_gen = n;
    return 0;
}
