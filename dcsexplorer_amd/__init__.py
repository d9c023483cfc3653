"""dcsexplorer_amd -- MI355X-native batched decoder for DCS pinball audio frames.

Python is only the thin test/bench harness over the C ABI of ``libdcs_hip.so``
(``include/dcs_hip.h``); the product is the shared library: host C++ (index pass, mixing
parameters, chunk planner, DCSDecoderHIP class) plus hand-written gfx950 HIP kernels.
"""
from .api import (  # noqa: F401
    OS93A, OS93B, OS94, OS95,
    FMT_93_T0, FMT_93B_T1, FMT_93A_T1, FMT_94_T0, FMT_94_T1_S0, FMT_94_T1_S3,
    FRAME_SAMPLES, FRAME_STOP, FRAME_FATAL, PREV_NONE, PREV_EXT, XFORM_93, XFORM_94,
    SRC_DTYPE, JOB_DTYPE, INDEX_DTYPE, IDX_SERIAL,
    DcsError, lib_path, load_library, build_id, lib_sha256,
    index_stream, index_streams, pack_streams, stream_params, volume_multiplier, mixing_multiplier, frame_scale,
    synth_stream, wav_header, dcsa_header, dcsa_parse, frame_diff, build_stream_batch, device_count, plan_chunks, pack_chunks, format_os,
    Context, Batch, RomSet, Sequencer, HW_DCS93, HW_DCS95,
    host_threads, partition_streams, decode_streams_sharded, Pipeline, make_refs, FRAME_TAIL_LOST, Node, DevicePath, node_cache_release, device_numa_node, bind_process_to_device_numa,
)
