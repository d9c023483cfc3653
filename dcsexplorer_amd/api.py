"""ctypes binding of include/dcs_hip.h.  Fails loudly when the shared library is missing: there is
no Python or CPU implementation of the decode path in this package."""
import ctypes
import os
import weakref

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

OS93A, OS93B, OS94, OS95 = 0, 1, 2, 3
ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_NO_MEMORY, ERR_CAPACITY, ERR_BAD_STREAM = -1, -2, -3, -4, -5, -6
FMT_93_T0, FMT_93B_T1, FMT_93A_T1, FMT_94_T0, FMT_94_T1_S0, FMT_94_T1_S3 = range(6)
FRAME_SAMPLES = 240
FRAME_STOP, FRAME_FATAL, FRAME_TAIL_LOST = 1, 2, 4
PREV_NONE = 0xFFFFFFFF
PREV_EXT = 0x80000000
XFORM_93, XFORM_94 = 0, 1

# numpy views of the ABI structs (layouts asserted against the library at load time)
SPLIT_DTYPE = np.dtype([("bitDelta", "<u2"), ("prv", "<u2"), ("prvDelta", "<u2"), ("state", "<u2")])
INDEX_DTYPE = np.dtype([("bitOff", "<u4"), ("nBits", "<u2"), ("hdrBits", "<u2"), ("bandType", "u1", (16,)),
                        ("preAdj", "<u2"), ("nBands", "u1"), ("flags", "u1"), ("split", SPLIT_DTYPE, (15,))])
SRC_DTYPE = np.dtype([("streamOff", "<u8"), ("mixMul", "<u2"), ("format", "u1"), ("hdrLen", "u1"),
                      ("idx", INDEX_DTYPE)])
JOB_DTYPE = np.dtype([("firstSrc", "<u4"), ("nSrc", "u1"), ("volShift", "u1"), ("xform", "u1"), ("flags", "u1"),
                      ("prev", "<u4"), ("reserved", "<u4")])
assert SRC_DTYPE.itemsize == 160 and JOB_DTYPE.itemsize == 16 and INDEX_DTYPE.itemsize == 148
IDX_SERIAL = 1


class StreamInfo(ctypes.Structure):
    _fields_ = [("nFrames", ctypes.c_int32), ("nBytes", ctypes.c_int32), ("formatType", ctypes.c_int32),
                ("formatSubType", ctypes.c_int32), ("header", ctypes.c_uint8 * 16), ("format", ctypes.c_int32),
                ("hdrLen", ctypes.c_int32), ("nValidFrames", ctypes.c_int32), ("payloadBits", ctypes.c_uint32)]


class StreamRef(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("len", ctypes.c_size_t), ("os", ctypes.c_int32),
                ("volume", ctypes.c_int32), ("level", ctypes.c_int32), ("channelVolume", ctypes.c_int32)]


LOC_DTYPE = np.dtype([("off", "<u8"), ("len", "<u4"), ("os", "<i4"), ("firstRecord", "<u8")])
INFO_DTYPE = np.dtype([("nFrames", "<i4"), ("nBytes", "<i4"), ("formatType", "<i4"), ("formatSubType", "<i4"),
                       ("header", "u1", (16,)), ("format", "<i4"), ("hdrLen", "<i4"), ("nValidFrames", "<i4"),
                       ("payloadBits", "<u4")])


class RomCheck(ctypes.Structure):
    _fields_ = [("status", ctypes.c_int32), ("hw", ctypes.c_int32), ("os", ctypes.c_int32),
                ("nominalVersion", ctypes.c_uint32), ("catalogOffset", ctypes.c_uint32), ("nTracks", ctypes.c_uint32),
                ("signature", ctypes.c_char * 128)]


class TrackInfo(ctypes.Structure):
    _fields_ = [("address", ctypes.c_uint32), ("channel", ctypes.c_int32), ("type", ctypes.c_int32),
                ("deferCode", ctypes.c_int32), ("time", ctypes.c_uint32), ("looping", ctypes.c_int32)]


TRACKOP_DTYPE = np.dtype([("offset", "<i4"), ("nestingLevel", "<i4"), ("loopParent", "<i4"), ("delayCount", "<u2"),
                          ("opcode", "u1"), ("nOperandBytes", "u1"), ("operandBytes", "u1", (8,))])
EXTRACT_DTYPE = np.dtype([("track", "<u4"), ("streamNum", "<u4"), ("address", "<u4"), ("level", "<i4")])
HW_DCS93, HW_DCS95 = 2, 3


class PipelineResult(ctypes.Structure):
    _fields_ = [("pcm", ctypes.c_void_p), ("err", ctypes.c_void_p), ("frameOffsets", ctypes.c_void_p),
                ("nFrames", ctypes.c_uint32), ("nStreams", ctypes.c_uint32), ("status", ctypes.c_int32),
                ("hostMs", ctypes.c_float), ("deviceMs", ctypes.c_float), ("path", ctypes.c_uint32)]


class DevicePathTimes(ctypes.Structure):
    _fields_ = [("passMs", ctypes.c_float), ("indexMs", ctypes.c_float), ("planMs", ctypes.c_float), ("packMs", ctypes.c_float),
                ("decodeMs", ctypes.c_float), ("planFlags", ctypes.c_uint32), ("nStreams", ctypes.c_uint32), ("nFrames", ctypes.c_uint32),
                ("framesPerWave", ctypes.c_uint32), ("algorithmicBytes", ctypes.c_uint64)]


class SynthParams(ctypes.Structure):
    _fields_ = [("seed", ctypes.c_uint64), ("format", ctypes.c_int32), ("nFrames", ctypes.c_int32),
                ("nBands", ctypes.c_int32), ("strideFromBand", ctypes.c_int32), ("profile", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


def _view(ptr, ctype, dtype, shape):
    """a numpy view of library memory (np.ctypeslib.as_array costs 0.6 ms per call for a 15 M-element array)"""
    addr = ptr.value if isinstance(ptr, ctypes.c_void_p) else ptr
    n = 1
    for d in shape:
        n *= int(d)
    if n == 0 or not addr:
        return np.zeros(shape, dtype=dtype)
    return np.frombuffer((ctype * n).from_address(addr), dtype=dtype).reshape(shape)


class DcsError(RuntimeError):
    def __init__(self, status, msg=""):
        super().__init__("libdcs_hip status %d %s" % (status, msg))
        self.status = status


_LIB = None
ABI_VERSION = 9                 # include/dcs_hip.h DCS_ABI_VERSION these bindings are written for

EXPORTS = [
    "dcs_abi_version", "dcs_build_id", "dcs_index_stream", "dcs_volume_multiplier", "dcs_mixing_multiplier", "dcs_frame_scale",
    "dcs_stream_params", "dcs_ctx_create", "dcs_ctx_destroy", "dcs_last_error", "dcs_device_count", "dcs_runtime_defaults", "dcs_ctx_set_batch_tails", "dcs_batch_package_bytes",
    "dcs_ctx_set_frames_per_wave", "dcs_ctx_set_tail_handoff", "dcs_ctx_set_large_list_path", "dcs_ctx_set_concurrent_batches", "dcs_ctx_set_cache_limits", "dcs_ctx_trim_cache", "dcs_ctx_cache_bytes", "dcs_plan_chunks2", "dcs_decode_batch", "dcs_batch_create", "dcs_batch_destroy", "dcs_batch_run", "dcs_batch_run_many", "dcs_pack_chunks",
    "dcs_batch_time", "dcs_batch_time_rotating", "dcs_batch_sync", "dcs_batch_download", "dcs_batch_download_view", "dcs_batch_device_pcm",
    "dcs_batch_algorithmic_bytes", "dcs_batch_num_jobs", "dcs_decode_streams", "dcs_count_stream_frames",
    "dcs_synth_stream", "dcs_plan_chunks", "dcs_index_streams", "dcs_index_streams_gpu",
    "dcs_index_streams_gpu_time", "dcs_stream_params_from", "dcs_decode_stream_sequence", "dcs_wav_header",
    "dcs_dcsa_header", "dcs_dcsa_parse", "dcs_write_wav", "dcs_write_dcsa", "dcs_frame_diff",
    "dcs_romset_create", "dcs_romset_destroy", "dcs_romset_last_error", "dcs_romset_add_rom", "dcs_romset_load_zip",
    "dcs_romset_load_zip_memory", "dcs_romset_check", "dcs_romset_set_version", "dcs_romset_num_tracks",
    "dcs_romset_pointer", "dcs_romset_bytes_behind", "dcs_romset_track_info", "dcs_romset_decompile", "dcs_romset_list_streams",
    "dcs_romset_extract_plan", "dcs_romset_stream_refs", "dcs_romset_extract_tracks_plan", "dcs_extract_tracks",
    "dcs_seq_create", "dcs_seq_destroy", "dcs_seq_last_error", "dcs_seq_set_master_volume", "dcs_seq_set_reported_version",
    "dcs_seq_write_data_port", "dcs_seq_add_track_command", "dcs_seq_clear_tracks", "dcs_seq_load_audio_stream",
    "dcs_seq_plan", "dcs_seq_pending_ticks", "dcs_seq_is_fatal", "dcs_seq_host_bytes", "dcs_seq_decode",
    "dcs_seq_create_standalone", "dcs_seq_load_audio_stream_mem", "dcs_seq_rewind", "dcs_seq_set_rewindable",
    "dcs_seq_tick", "dcs_seq_fatal_tick", "dcs_seq_stream_playing",
    "dcs_decode_batch_live", "dcs_seq_decode_view", "dcs_seq_plan_ahead", "dcs_seq_stream_playing_at", "dcs_seq_tracks_active_at", "dcs_ctx_call_floor",
    "dcs_host_threads", "dcs_partition_streams", "dcs_decode_streams_sharded",
    "dcs_ctx_set_frames_per_chunk", "dcs_index_stream_literal", "dcs_pack_chunks_device", "dcs_batch_abi_bytes", "dcs_batch_num_chunks", "dcs_batch_frames_per_wave", "dcs_ctx_clock_mhz", "dcs_ctx_link_rate", "dcs_ctx_set_test_hooks",
    "dcs_pipeline_create", "dcs_pipeline_destroy", "dcs_pipeline_submit", "dcs_pipeline_collect",
    "dcs_device_path_create", "dcs_device_path_run", "dcs_device_path_run_many", "dcs_device_path_download", "dcs_device_path_destroy",
    "dcs_node_create", "dcs_node_destroy", "dcs_node_submit", "dcs_node_collect", "dcs_node_num_devices", "dcs_node_device_info",
    "dcs_node_last_error", "dcs_node_cache_release", "dcs_device_numa_node",
]


def build_id():
    """the digest of sources and flags the LOADED library was built from (dcs_build_id)"""
    return load_library().dcs_build_id().decode()


def lib_sha256():
    import hashlib
    return hashlib.sha256(open(lib_path(), "rb").read()).hexdigest()


def lib_path():
    # DCS_HIP_LIB: another build of the same library (kernel A/B comparisons, tools/ab.sh); never a different backend
    return os.environ.get("DCS_HIP_LIB") or os.path.join(HERE, "libdcs_hip.so")


def load_library():
    """Load libdcs_hip.so.  If torch is importable it is imported first so that the process uses a
    single HIP runtime (torch bundles its own libamdhip64 with the same soname)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C dcsexplorer_amd/csrc` (there is no fallback implementation)" % path)
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    L = ctypes.CDLL(path)
    vp, u32, i32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_size_t
    L.dcs_abi_version.restype = u32
    L.dcs_build_id.restype = ctypes.c_char_p
    if L.dcs_abi_version() != ABI_VERSION:
        # (DCS_HIP_LIB may name another BUILD of this library, tools/ab.sh; one with other struct layouts must not be driven
        # through these bindings: dcs_pipeline_collect writes sizeof(DcsPipelineResult) bytes into the caller's struct)
        raise ImportError("%s has ABI version %d, these bindings are written for %d: rebuild it (make -C dcsexplorer_amd/csrc)"
                          % (path, L.dcs_abi_version(), ABI_VERSION))
    L.dcs_index_stream.restype = i32
    L.dcs_index_stream.argtypes = [i32, vp, sz, vp, u32, ctypes.POINTER(StreamInfo)]
    L.dcs_index_stream_literal.restype = i32
    L.dcs_index_stream_literal.argtypes = [i32, vp, sz, vp, u32, ctypes.POINTER(StreamInfo)]
    L.dcs_volume_multiplier.restype = ctypes.c_uint16
    L.dcs_volume_multiplier.argtypes = [ctypes.c_int]
    L.dcs_mixing_multiplier.restype = ctypes.c_uint16
    L.dcs_mixing_multiplier.argtypes = [i32, ctypes.c_int, ctypes.c_int]
    L.dcs_frame_scale.restype = ctypes.c_int
    L.dcs_frame_scale.argtypes = [ctypes.c_uint16, vp, vp, ctypes.c_int]
    L.dcs_stream_params.restype = i32
    L.dcs_stream_params.argtypes = [i32, ctypes.c_int, ctypes.c_int, ctypes.c_int, u32, vp, vp]
    L.dcs_ctx_create.restype = i32
    L.dcs_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.dcs_ctx_destroy.restype = None
    L.dcs_ctx_destroy.argtypes = [vp]
    L.dcs_last_error.restype = ctypes.c_char_p
    L.dcs_last_error.argtypes = [vp]
    L.dcs_device_count.restype = ctypes.c_int
    L.dcs_ctx_set_frames_per_wave.restype = i32
    L.dcs_ctx_set_frames_per_wave.argtypes = [vp, ctypes.c_int]
    L.dcs_ctx_set_tail_handoff.restype = i32
    L.dcs_ctx_set_tail_handoff.argtypes = [vp, ctypes.c_int]
    L.dcs_ctx_set_large_list_path.restype = i32
    L.dcs_ctx_set_large_list_path.argtypes = [vp, ctypes.c_int]
    L.dcs_ctx_set_concurrent_batches.restype = i32
    L.dcs_ctx_set_concurrent_batches.argtypes = [vp, ctypes.c_int]
    u64p = ctypes.POINTER(ctypes.c_uint64)
    L.dcs_ctx_set_cache_limits.restype = i32
    L.dcs_ctx_set_cache_limits.argtypes = [vp, ctypes.c_uint64, ctypes.c_uint64]
    L.dcs_ctx_trim_cache.restype = i32
    L.dcs_ctx_trim_cache.argtypes = [vp, u64p, u64p]
    L.dcs_ctx_cache_bytes.restype = i32
    L.dcs_ctx_cache_bytes.argtypes = [vp, u64p, u64p, u64p, u64p]
    L.dcs_pack_chunks.restype = i32
    L.dcs_pack_chunks.argtypes = [vp, u32, vp, vp, sz, ctypes.c_int, vp, sz, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.dcs_plan_chunks2.restype = i32
    L.dcs_plan_chunks2.argtypes = [vp, u32, vp, ctypes.c_int, ctypes.c_int, vp, sz, ctypes.POINTER(u32)]
    L.dcs_decode_batch.restype = i32
    L.dcs_decode_batch.argtypes = [vp, vp, sz, vp, u32, vp, u32, vp, u32, vp, vp, vp]
    L.dcs_batch_create.restype = i32
    L.dcs_batch_create.argtypes = [vp, vp, sz, vp, u32, vp, u32, vp, u32, ctypes.POINTER(vp)]
    L.dcs_batch_destroy.restype = None
    L.dcs_batch_destroy.argtypes = [vp]
    L.dcs_batch_run.restype = i32
    L.dcs_batch_run.argtypes = [vp, vp]
    L.dcs_batch_run_many.restype = i32
    L.dcs_batch_run_many.argtypes = [vp, vp, ctypes.c_int]
    L.dcs_batch_time.restype = i32
    L.dcs_batch_time.argtypes = [vp, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    L.dcs_batch_time_rotating.restype = i32
    L.dcs_batch_time_rotating.argtypes = [ctypes.POINTER(vp), u32, vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    L.dcs_batch_sync.restype = i32
    L.dcs_batch_sync.argtypes = [vp]
    L.dcs_batch_download.restype = i32
    L.dcs_batch_download.argtypes = [vp, vp, vp, vp]
    L.dcs_batch_download_view.restype = i32
    L.dcs_batch_download_view.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.dcs_batch_device_pcm.restype = vp
    L.dcs_batch_device_pcm.argtypes = [vp]
    L.dcs_batch_algorithmic_bytes.restype = ctypes.c_uint64
    L.dcs_batch_algorithmic_bytes.argtypes = [vp]
    L.dcs_batch_num_jobs.restype = u32
    L.dcs_batch_num_jobs.argtypes = [vp]
    L.dcs_decode_streams.restype = i32
    L.dcs_decode_streams.argtypes = [vp, vp, u32, u32, vp, sz, vp, vp]
    L.dcs_count_stream_frames.restype = i32
    L.dcs_count_stream_frames.argtypes = [vp, u32, u32, ctypes.POINTER(ctypes.c_uint64)]
    L.dcs_synth_stream.restype = i32
    L.dcs_synth_stream.argtypes = [ctypes.POINTER(SynthParams), vp, sz, ctypes.POINTER(sz)]
    L.dcs_plan_chunks.restype = i32
    L.dcs_plan_chunks.argtypes = [vp, u32, vp, ctypes.c_int, vp, sz, ctypes.POINTER(u32)]
    L.dcs_stream_params_from.restype = i32
    L.dcs_stream_params_from.argtypes = [i32, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_uint16, u32, vp, vp]
    L.dcs_decode_stream_sequence.restype = i32
    L.dcs_decode_stream_sequence.argtypes = [vp, vp, u32, u32, vp, sz, vp, vp]
    L.dcs_wav_header.restype = None
    L.dcs_wav_header.argtypes = [u32, vp]
    L.dcs_dcsa_header.restype = i32
    L.dcs_dcsa_header.argtypes = [i32, u32, vp]
    L.dcs_dcsa_parse.restype = i32
    L.dcs_dcsa_parse.argtypes = [vp, sz, ctypes.POINTER(i32), ctypes.POINTER(vp), ctypes.POINTER(u32)]
    L.dcs_write_wav.restype = i32
    L.dcs_write_wav.argtypes = [ctypes.c_char_p, vp, u32]
    L.dcs_write_dcsa.restype = i32
    L.dcs_write_dcsa.argtypes = [ctypes.c_char_p, i32, vp, u32]
    L.dcs_frame_diff.restype = ctypes.c_int
    L.dcs_frame_diff.argtypes = [ctypes.c_uint64, vp, vp, ctypes.c_char_p, sz, ctypes.POINTER(sz)]
    L.dcs_romset_create.restype = vp
    L.dcs_romset_create.argtypes = []
    L.dcs_romset_destroy.restype = None
    L.dcs_romset_destroy.argtypes = [vp]
    L.dcs_romset_last_error.restype = ctypes.c_char_p
    L.dcs_romset_last_error.argtypes = [vp]
    L.dcs_romset_add_rom.restype = i32
    L.dcs_romset_add_rom.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, sz]
    L.dcs_romset_load_zip.restype = i32
    L.dcs_romset_load_zip.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p]
    L.dcs_romset_load_zip_memory.restype = i32
    L.dcs_romset_load_zip_memory.argtypes = [vp, ctypes.c_char_p, sz, ctypes.c_char_p, ctypes.c_char_p]
    L.dcs_romset_check.restype = i32
    L.dcs_romset_check.argtypes = [vp, ctypes.POINTER(RomCheck)]
    L.dcs_romset_set_version.restype = i32
    L.dcs_romset_set_version.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    L.dcs_romset_num_tracks.restype = u32
    L.dcs_romset_num_tracks.argtypes = [vp]
    L.dcs_romset_pointer.restype = i32
    L.dcs_romset_pointer.argtypes = [vp, u32, ctypes.POINTER(vp), ctypes.POINTER(sz), ctypes.POINTER(ctypes.c_int)]
    L.dcs_romset_track_info.restype = i32
    L.dcs_romset_track_info.argtypes = [vp, u32, ctypes.POINTER(TrackInfo)]
    L.dcs_romset_decompile.restype = i32
    L.dcs_romset_decompile.argtypes = [vp, u32, vp, u32, ctypes.POINTER(u32)]
    L.dcs_romset_list_streams.restype = i32
    L.dcs_romset_list_streams.argtypes = [vp, vp, u32, ctypes.POINTER(u32)]
    L.dcs_romset_extract_plan.restype = i32
    L.dcs_romset_extract_plan.argtypes = [vp, vp, u32, ctypes.POINTER(u32)]
    L.dcs_romset_extract_tracks_plan.restype = i32
    L.dcs_romset_extract_tracks_plan.argtypes = [vp, vp, u32, ctypes.POINTER(u32)]
    L.dcs_extract_tracks.restype = i32
    L.dcs_extract_tracks.argtypes = [vp, vp, vp, u32, vp, sz, vp, vp]
    L.dcs_romset_stream_refs.restype = i32
    L.dcs_romset_stream_refs.argtypes = [vp, vp, u32, ctypes.c_int, vp]
    L.dcs_seq_create.restype = vp
    L.dcs_seq_create.argtypes = [vp]
    L.dcs_seq_destroy.restype = None
    L.dcs_seq_destroy.argtypes = [vp]
    L.dcs_seq_last_error.restype = ctypes.c_char_p
    L.dcs_seq_last_error.argtypes = [vp]
    for name, args in (("dcs_seq_set_master_volume", [vp, ctypes.c_int]), ("dcs_seq_set_reported_version", [vp, ctypes.c_uint16]),
                       ("dcs_seq_write_data_port", [vp, ctypes.c_uint8]), ("dcs_seq_add_track_command", [vp, ctypes.c_uint16]),
                       ("dcs_seq_clear_tracks", [vp]), ("dcs_seq_load_audio_stream", [vp, ctypes.c_int, u32, ctypes.c_int]),
                       ("dcs_seq_plan", [vp, u32]), ("dcs_seq_decode", [vp, vp, vp, sz, vp])):
        getattr(L, name).restype = i32
        getattr(L, name).argtypes = args
    L.dcs_seq_create_standalone.restype = vp
    L.dcs_seq_create_standalone.argtypes = [i32]
    L.dcs_seq_load_audio_stream_mem.restype = i32
    L.dcs_seq_load_audio_stream_mem.argtypes = [vp, ctypes.c_int, ctypes.c_char_p, sz, ctypes.c_int]
    L.dcs_seq_rewind.restype = i32
    L.dcs_seq_rewind.argtypes = [vp, u32]
    L.dcs_seq_plan_ahead.restype = i32
    L.dcs_seq_plan_ahead.argtypes = [vp, u32, u32, ctypes.POINTER(u32)]
    L.dcs_seq_decode_view.restype = i32
    L.dcs_seq_decode_view.argtypes = [vp, vp, ctypes.POINTER(vp), ctypes.POINTER(u32), ctypes.POINTER(vp)]
    L.dcs_seq_stream_playing_at.restype = ctypes.c_int
    L.dcs_seq_stream_playing_at.argtypes = [vp, u32, ctypes.c_int]
    L.dcs_seq_stream_playing.restype = ctypes.c_int
    L.dcs_seq_stream_playing.argtypes = [vp, ctypes.c_int]
    L.dcs_seq_tracks_active_at.restype = ctypes.c_int
    L.dcs_seq_tracks_active_at.argtypes = [vp, u32]
    L.dcs_ctx_call_floor.restype = i32
    L.dcs_ctx_call_floor.argtypes = [vp, u32, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float)]
    L.dcs_decode_batch_live.restype = i32
    L.dcs_decode_batch_live.argtypes = [vp, vp, sz, ctypes.c_uint64, vp, u32, vp, u32, vp, u32, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp)]
    L.dcs_seq_set_rewindable.restype = i32
    L.dcs_seq_set_rewindable.argtypes = [vp, ctypes.c_int]
    L.dcs_seq_pending_ticks.restype = u32
    L.dcs_seq_pending_ticks.argtypes = [vp]
    L.dcs_seq_is_fatal.restype = ctypes.c_int
    L.dcs_seq_is_fatal.argtypes = [vp]
    L.dcs_seq_host_bytes.restype = u32
    L.dcs_seq_host_bytes.argtypes = [vp, vp, u32]
    L.dcs_index_streams.restype = i32
    L.dcs_index_streams.argtypes = [vp, u32, ctypes.c_int, vp, vp, vp]
    L.dcs_index_streams_gpu.restype = i32
    L.dcs_index_streams_gpu.argtypes = [vp, vp, sz, vp, u32, vp, ctypes.c_uint64, vp]
    L.dcs_index_streams_gpu_time.restype = i32
    L.dcs_index_streams_gpu_time.argtypes = [vp, ctypes.c_int, ctypes.POINTER(ctypes.c_float)]
    L.dcs_pack_chunks_device.restype = i32
    L.dcs_pack_chunks_device.argtypes = [vp, vp, u32, vp, u32, vp, sz, ctypes.c_int, vp, sz, ctypes.POINTER(u32), ctypes.POINTER(u32)]
    L.dcs_ctx_set_frames_per_chunk.restype = i32
    L.dcs_ctx_set_frames_per_chunk.argtypes = [vp, ctypes.c_int]
    L.dcs_batch_abi_bytes.restype = ctypes.c_uint64
    L.dcs_batch_abi_bytes.argtypes = [vp]
    L.dcs_batch_num_chunks.restype = u32
    L.dcs_batch_num_chunks.argtypes = [vp]
    L.dcs_batch_package_bytes.restype = u32
    L.dcs_batch_package_bytes.argtypes = [vp]
    L.dcs_ctx_set_batch_tails.restype = i32
    L.dcs_ctx_set_batch_tails.argtypes = [vp, ctypes.c_int]
    L.dcs_runtime_defaults.restype = ctypes.c_int
    L.dcs_runtime_defaults.argtypes = []
    L.dcs_batch_frames_per_wave.restype = ctypes.c_int
    L.dcs_batch_frames_per_wave.argtypes = [vp]
    L.dcs_ctx_clock_mhz.restype = i32
    L.dcs_ctx_clock_mhz.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.dcs_ctx_link_rate.restype = i32
    L.dcs_ctx_link_rate.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
    L.dcs_ctx_set_test_hooks.restype = i32
    L.dcs_ctx_set_test_hooks.argtypes = [vp, u32, ctypes.c_int]
    L.dcs_pipeline_create.restype = i32
    L.dcs_pipeline_create.argtypes = [vp, ctypes.c_int, u32, ctypes.POINTER(vp)]
    L.dcs_pipeline_destroy.restype = None
    L.dcs_pipeline_destroy.argtypes = [vp]
    L.dcs_pipeline_submit.restype = i32
    L.dcs_pipeline_submit.argtypes = [vp, vp, u32, u32]
    L.dcs_pipeline_collect.restype = i32
    L.dcs_pipeline_collect.argtypes = [vp, ctypes.POINTER(PipelineResult)]
    L.dcs_device_path_create.restype = i32
    L.dcs_device_path_create.argtypes = [vp, vp, u32, u32, ctypes.POINTER(vp)]
    L.dcs_device_path_run.restype = i32
    L.dcs_device_path_run.argtypes = [vp, ctypes.c_int, ctypes.POINTER(DevicePathTimes)]
    L.dcs_device_path_run_many.restype = i32
    L.dcs_device_path_run_many.argtypes = [vp, ctypes.c_int]
    L.dcs_device_path_download.restype = i32
    L.dcs_device_path_download.argtypes = [vp, vp, vp, vp]
    L.dcs_device_path_destroy.restype = None
    L.dcs_device_path_destroy.argtypes = [vp]
    L.dcs_node_create.restype = i32
    L.dcs_node_create.argtypes = [vp, u32, ctypes.c_int, u32, ctypes.POINTER(vp)]
    L.dcs_node_destroy.restype = None
    L.dcs_node_destroy.argtypes = [vp]
    L.dcs_node_submit.restype = i32
    L.dcs_node_submit.argtypes = [vp, vp, u32, u32]
    L.dcs_node_collect.restype = i32
    L.dcs_node_collect.argtypes = [vp, ctypes.POINTER(PipelineResult), ctypes.POINTER(ctypes.c_int)]
    L.dcs_node_num_devices.restype = u32
    L.dcs_node_num_devices.argtypes = [vp]
    L.dcs_node_device_info.restype = i32
    L.dcs_node_device_info.argtypes = [vp, u32, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_uint64)]
    L.dcs_node_last_error.restype = ctypes.c_char_p
    L.dcs_node_last_error.argtypes = [vp]
    L.dcs_node_cache_release.restype = None
    L.dcs_node_cache_release.argtypes = []
    L.dcs_device_numa_node.restype = ctypes.c_int
    L.dcs_device_numa_node.argtypes = [ctypes.c_int]
    L.dcs_host_threads.restype = ctypes.c_int
    L.dcs_host_threads.argtypes = []
    L.dcs_partition_streams.restype = i32
    L.dcs_partition_streams.argtypes = [vp, u32, u32, vp]
    L.dcs_decode_streams_sharded.restype = i32
    L.dcs_decode_streams_sharded.argtypes = [vp, u32, vp, u32, u32, vp, sz, vp, vp, vp]
    _LIB = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def _check(st, ctx=None):
    if st != 0:
        msg = load_library().dcs_last_error(ctx)
        raise DcsError(st, msg.decode() if msg else "")


# ---------------------------------------------------------------------------------------------- host side
def index_stream(os_, stream, literal=False):
    """dcs_index_stream (literal: dcs_index_stream_literal): returns (index records as INDEX_DTYPE array, StreamInfo)"""
    L = load_library()
    buf = np.frombuffer(bytes(stream), dtype=np.uint8)
    nframes = (int(buf[0]) << 8) | int(buf[1])
    out = np.zeros(max(nframes, 1), dtype=INDEX_DTYPE)
    info = StreamInfo()
    fn = L.dcs_index_stream_literal if literal else L.dcs_index_stream
    st = fn(os_, _ptr(buf), buf.size, _ptr(out), nframes, ctypes.byref(info))
    if st != 0:
        raise DcsError(st)
    return out[:info.nValidFrames], info


def _info_from_record(r):
    info = StreamInfo()
    ctypes.memmove(ctypes.byref(info), r.tobytes(), ctypes.sizeof(info))
    return info


def _frame_counts(streams):
    return np.array([(s[1][0] << 8) | s[1][1] for s in streams], dtype=np.uint64)


def _split_records(out, infos, first):
    return [(out[int(first[k]):int(first[k]) + int(infos[k]["nValidFrames"])], _info_from_record(infos[k]))
            for k in range(len(infos))]


def index_streams(streams, threads=0):
    """dcs_index_streams: index many (os, bytes, ...) streams on `threads` host threads (0 = all).
    Returns a list of (records, StreamInfo), one per stream, identical to index_stream on each."""
    L = load_library()
    streams = list(streams)
    if not streams:
        return []
    counts = _frame_counts(streams)
    first = np.concatenate([[0], np.cumsum(counts)]).astype(np.uint64)
    keep = [np.frombuffer(bytes(s[1]), dtype=np.uint8) for s in streams]
    refs = (StreamRef * len(streams))()
    for k, s in enumerate(streams):
        refs[k].data = keep[k].ctypes.data
        refs[k].len = keep[k].size
        refs[k].os = s[0]
    out = np.zeros(max(int(first[-1]), 1), dtype=INDEX_DTYPE)
    infos = np.zeros(len(streams), dtype=INFO_DTYPE)
    st = L.dcs_index_streams(refs, len(streams), threads, _ptr(out), _ptr(first), _ptr(infos))
    if st != 0:
        raise DcsError(st)
    return _split_records(out, infos, first)


def pack_streams(streams, pad=64):
    """Lay (os, bytes, ...) streams out in one dword-aligned blob: -> (blob bytes, LOC_DTYPE array)"""
    blob = bytearray()
    locs = np.zeros(len(streams), dtype=LOC_DTYPE)
    rec = 0
    for k, s in enumerate(streams):
        while len(blob) & 3:
            blob.append(0)
        data = bytes(s[1])
        locs[k] = (len(blob), len(data), s[0], rec)
        rec += (data[0] << 8) | data[1]
        blob += data
        blob += bytes(pad)
    return bytes(blob), locs


def volume_multiplier(vol):
    return load_library().dcs_volume_multiplier(vol)


def mixing_multiplier(os_, level_sum, channel_volume=0xFF):
    return load_library().dcs_mixing_multiplier(os_, level_sum, channel_volume)


def frame_scale(vol_mult, mix_muls, active=None):
    mm = np.ascontiguousarray(mix_muls, dtype=np.uint16).copy()
    act = None if active is None else np.ascontiguousarray(active, dtype=np.uint8)
    vs = load_library().dcs_frame_scale(vol_mult, _ptr(mm), _ptr(act), mm.size)
    return vs, mm


def stream_params(os_, volume, level, nframes, channel_volume=0xFF):
    mm = np.zeros(nframes, dtype=np.uint16)
    vs = np.zeros(nframes, dtype=np.uint8)
    _check(load_library().dcs_stream_params(os_, volume, level, channel_volume, nframes, _ptr(mm), _ptr(vs)))
    return mm, vs


def wav_header(nframes):
    out = np.zeros(44, dtype=np.uint8)
    load_library().dcs_wav_header(nframes, _ptr(out))
    return out.tobytes()


def dcsa_header(os_, nbytes):
    out = np.zeros(36, dtype=np.uint8)
    st = load_library().dcs_dcsa_header(os_, nbytes, _ptr(out))
    if st != 0:
        raise DcsError(st)
    return out.tobytes()


def dcsa_parse(data):
    """-> (os, stream bytes) of a "DCSa" container; raises DcsError(ERR_BAD_STREAM) if it is not one"""
    buf = np.frombuffer(bytes(data), dtype=np.uint8) if len(data) else np.zeros(1, dtype=np.uint8)
    os_, ptr, n = ctypes.c_int32(), ctypes.c_void_p(), ctypes.c_uint32()
    st = load_library().dcs_dcsa_parse(_ptr(buf), len(data), ctypes.byref(os_), ctypes.byref(ptr), ctypes.byref(n))
    if st != 0:
        raise DcsError(st)
    off = ptr.value - buf.ctypes.data
    return os_.value, bytes(data)[off:off + n.value]


def frame_diff(frame_no, mine, theirs):
    """-> (number of differing samples, the block the reference's --validate log would hold)"""
    a = np.ascontiguousarray(mine, dtype=np.int16)
    b = np.ascontiguousarray(theirs, dtype=np.int16)
    assert a.size == FRAME_SAMPLES and b.size == FRAME_SAMPLES
    text = ctypes.create_string_buffer(8192)
    n = ctypes.c_size_t()
    d = load_library().dcs_frame_diff(frame_no, _ptr(a), _ptr(b), text, len(text), ctypes.byref(n))
    return d, text.raw[:n.value].decode()


def synth_stream(fmt, nframes, seed, nbands=16, stride_from=16, profile=0):
    L = load_library()
    p = SynthParams(seed=seed, format=fmt, nFrames=nframes, nBands=nbands, strideFromBand=stride_from,
                    profile=profile, reserved=0)
    n = ctypes.c_size_t(0)
    _check(L.dcs_synth_stream(ctypes.byref(p), None, 0, ctypes.byref(n)))
    out = np.zeros(n.value, dtype=np.uint8)
    _check(L.dcs_synth_stream(ctypes.byref(p), _ptr(out), out.size, ctypes.byref(n)))
    return out.tobytes()


def format_os(fmt, prefer_95=False, prefer_93a=False):
    """an OS version whose decoder parses `fmt`"""
    if fmt == FMT_93A_T1:
        return OS93A
    if fmt == FMT_93B_T1:
        return OS93B
    if fmt == FMT_93_T0:
        return OS93A if prefer_93a else OS93B
    return OS95 if prefer_95 else OS94


def build_stream_batch(streams, extra_frames=0, pad=64, indexer=None):
    """Host-side batch description for independent streams, each played alone from a fresh decoder
    (LoadAudioStream(0, ptr, level), DCSDecoderNative.cpp:1387).

    streams: iterable of (os, bytes, volume, level).  indexer: None (dcs_index_stream per stream), or a
    callable streams -> [(records, StreamInfo)] such as index_streams or Context.index_streams_gpu.
    Returns dict(blob, srcs, jobs, first_job)."""
    blob = bytearray()
    srcs, jobs, first = [], [], []
    njobs = 0
    streams = list(streams)
    indexed = indexer(streams) if indexer is not None else None
    for k, (os_, data, volume, level) in enumerate(streams):
        idx, info = indexed[k] if indexed is not None else index_stream(os_, data)
        nframes = info.nFrames
        mm, vs = stream_params(os_, volume, level, nframes)
        while len(blob) & 3:
            blob.append(0)
        off = len(blob)
        blob += bytes(data)
        blob += bytes(max(pad, info.nBytes - len(data) + 8))     # bytes past a (damaged) stream's end read as zero
        nvalid = info.nValidFrames
        s = np.zeros(nvalid, dtype=SRC_DTYPE)
        s["streamOff"] = off
        s["mixMul"] = mm[:nvalid]
        s["format"] = info.format
        s["hdrLen"] = info.hdrLen
        s["idx"] = idx
        total = nframes + extra_frames
        j = np.zeros(total, dtype=JOB_DTYPE)
        nsrc_before = sum(len(x) for x in srcs)
        j["firstSrc"][:nvalid] = nsrc_before + np.arange(nvalid)
        j["nSrc"][:nvalid] = 1
        j["volShift"][:nvalid] = vs[:nvalid]
        j["volShift"][nvalid:] = 8
        j["xform"] = XFORM_93 if os_ in (OS93A, OS93B) else XFORM_94
        j["prev"] = njobs + np.arange(total, dtype=np.int64) - 1
        j["prev"][0] = PREV_NONE
        srcs.append(s)
        jobs.append(j)
        first.append(njobs)
        njobs += total
    first.append(njobs)
    return dict(blob=bytes(blob), srcs=np.concatenate(srcs) if srcs else np.zeros(0, SRC_DTYPE),
                jobs=np.concatenate(jobs), first_job=np.array(first, dtype=np.int64))


def plan_chunks(jobs, fpw, srcs=None, handoff=True):
    """dcs_plan_chunks2 -> slots [nChunks, fpw] as a structured array (job, prevSlot, flags)"""
    L = load_library()
    jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
    srcs = None if srcs is None else np.ascontiguousarray(srcs, dtype=SRC_DTYPE)
    n = ctypes.c_uint32(0)
    _check(L.dcs_plan_chunks2(_ptr(jobs), jobs.size, _ptr(srcs), fpw, int(handoff), None, 0, ctypes.byref(n)))
    raw = np.zeros(n.value * fpw, dtype=np.uint64)
    _check(L.dcs_plan_chunks2(_ptr(jobs), jobs.size, _ptr(srcs), fpw, int(handoff), _ptr(raw), raw.size, ctypes.byref(n)))
    out = np.zeros(raw.size, dtype=[("job", "<u4"), ("prevSlot", "u1"), ("flags", "u1")])
    out["job"] = raw & 0xFFFFFFFF
    out["prevSlot"] = (raw >> 32) & 0xFF
    out["flags"] = (raw >> 40) & 0xFF
    return out.reshape(n.value, fpw)


def pack_chunks(blob, srcs, jobs, fpw):
    """dcs_pack_chunks -> uint8 array [nChunks, packageBytes] (what a batch uploads for unpack round 0)"""
    L = load_library()
    blob_a = np.frombuffer(bytes(blob), dtype=np.uint8)
    srcs = np.ascontiguousarray(srcs, dtype=SRC_DTYPE)
    jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
    n, pb = ctypes.c_uint32(0), ctypes.c_uint32(0)
    _check(L.dcs_pack_chunks(_ptr(jobs), jobs.size, _ptr(srcs), _ptr(blob_a), blob_a.size, fpw, None, 0, ctypes.byref(n), ctypes.byref(pb)))
    out = np.zeros((n.value, pb.value), dtype=np.uint8)
    _check(L.dcs_pack_chunks(_ptr(jobs), jobs.size, _ptr(srcs), _ptr(blob_a), blob_a.size, fpw, _ptr(out), out.size, ctypes.byref(n), ctypes.byref(pb)))
    return out


def device_count():
    return load_library().dcs_device_count()


def host_threads():
    """dcs_host_threads: CPUs this process can really use (affinity mask, cgroup quota)"""
    return load_library().dcs_host_threads()


def partition_streams(frame_counts, n_parts):
    """dcs_partition_streams: cut points (n_parts + 1) of contiguous ranges balanced by total frame count"""
    fc = np.ascontiguousarray(frame_counts, dtype=np.uint32)
    cut = np.zeros(n_parts + 1, dtype=np.uint32)
    st = load_library().dcs_partition_streams(_ptr(fc), fc.size, n_parts, _ptr(cut))
    if st != 0:
        raise DcsError(st)
    return cut


def _stream_refs(streams):
    keep = [np.frombuffer(bytes(s[1]), dtype=np.uint8) for s in streams]
    refs = (StreamRef * len(streams))()
    for k, s in enumerate(streams):
        refs[k].data = keep[k].ctypes.data
        refs[k].len = keep[k].size
        refs[k].os = s[0]
        refs[k].volume = s[2]
        refs[k].level = s[3]
        refs[k].channelVolume = 0xFF
    return refs, keep


def decode_streams_sharded(device_ids, streams, extra_frames=0):
    """dcs_decode_streams_sharded: (os, bytes, volume, level) streams over several devices (one host thread and one
    context per device, range partition balanced by frames) -> (pcm [frames, 240], err, first frame of each stream,
    first stream of each device)"""
    L = load_library()
    streams = list(streams)
    refs, keep = _stream_refs(streams)
    total = int(sum(((int(k[0]) << 8) | int(k[1])) + extra_frames for k in keep))
    pcm = np.zeros((total, FRAME_SAMPLES), dtype=np.int16)
    err = np.zeros(total, dtype=np.uint32)
    first = np.zeros(len(streams) + 1, dtype=np.uint32)
    devs = np.ascontiguousarray(device_ids, dtype=np.int32)
    cut = np.zeros(devs.size + 1, dtype=np.uint32)
    st = L.dcs_decode_streams_sharded(_ptr(devs), devs.size, refs, len(streams), extra_frames, _ptr(pcm), total,
                                      _ptr(first), _ptr(err), _ptr(cut))
    if st != 0:
        msg = L.dcs_last_error(None)
        raise DcsError(st, msg.decode() if msg else "")
    return pcm, err, first, cut


# ---------------------------------------------------------------------------------------------- device side
class Context:
    """DcsCtx: one GPU.  Raises DcsError(DCS_ERR_NO_DEVICE) when there is no gfx950 device."""

    def __init__(self, device=0):
        self.L = load_library()
        h = ctypes.c_void_p()
        st = self.L.dcs_ctx_create(device, ctypes.byref(h))
        if st != 0:
            msg = self.L.dcs_last_error(None)
            raise DcsError(st, msg.decode() if msg else "")
        self.h = h
        self.device = int(device)
        self._batches = weakref.WeakSet()       # a DcsBatch must be destroyed before its DcsCtx (dcs_hip.h)

    def close(self):
        if self.h:
            for b in list(self._batches):
                b.close()
            self.L.dcs_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def link_rate(self):
        """GB/s, device memory -> pinned host memory, measured now (dcs_ctx_link_rate)"""
        v = ctypes.c_float(0)
        _check(self.L.dcs_ctx_link_rate(self.h, ctypes.byref(v)), self.h)
        return float(v.value)

    def set_concurrent_batches(self, enable=True):
        """batches created afterwards may run next to other decode launches on the GPU (dcs_ctx_set_concurrent_batches)"""
        _check(self.L.dcs_ctx_set_concurrent_batches(self.h, int(bool(enable))), self.h)

    def set_batch_tails(self, all_frames=True):
        """resident batches created afterwards store every frame's tail (True) or the last frame's of every chain (False, default)"""
        _check(self.L.dcs_ctx_set_batch_tails(self.h, int(bool(all_frames))), self.h)

    def set_cache_limits(self, device_bytes, pinned_bytes):
        _check(self.L.dcs_ctx_set_cache_limits(self.h, int(device_bytes), int(pinned_bytes)), self.h)

    def trim_cache(self):
        """release every cached buffer -> (device bytes, pinned bytes) released"""
        d, p = ctypes.c_uint64(0), ctypes.c_uint64(0)
        _check(self.L.dcs_ctx_trim_cache(self.h, ctypes.byref(d), ctypes.byref(p)), self.h)
        return d.value, p.value

    def cache_bytes(self):
        """-> (device bytes cached, pinned bytes cached, device limit, pinned limit)"""
        v = [ctypes.c_uint64(0) for _ in range(4)]
        _check(self.L.dcs_ctx_cache_bytes(self.h, *[ctypes.byref(x) for x in v]), self.h)
        return tuple(x.value for x in v)

    def set_frames_per_wave(self, fpw):
        _check(self.L.dcs_ctx_set_frames_per_wave(self.h, fpw), self.h)

    def set_frames_per_chunk(self, frames):
        _check(self.L.dcs_ctx_set_frames_per_chunk(self.h, frames), self.h)

    def set_tail_handoff(self, enable):
        _check(self.L.dcs_ctx_set_tail_handoff(self.h, int(bool(enable))), self.h)

    def set_large_list_path(self, on_device, shared=True):
        """dcs_decode_streams on a large list: 0 = index pass on the host's pool; 1 = index walk, planner and packer on the device;
        2 (default) = the device path with the host pool walking the list's first parts next to the index kernel.
        on_device may also be the mode itself (0, 1, 2)."""
        mode = int(on_device) if on_device in (0, 1, 2) and not isinstance(on_device, bool) else (0 if not on_device else (2 if shared else 1))
        _check(self.L.dcs_ctx_set_large_list_path(self.h, mode), self.h)

    def decode_batch(self, blob, srcs, jobs, tails_in=None, want_tails=False):
        blob_a = np.frombuffer(bytes(blob), dtype=np.uint8)
        srcs = np.ascontiguousarray(srcs, dtype=SRC_DTYPE)
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        n = jobs.size
        pcm = np.zeros((n, FRAME_SAMPLES), dtype=np.int16)
        err = np.zeros(n, dtype=np.uint32)
        tails = np.zeros((n, 16), dtype=np.int16) if want_tails else None
        tin = None if tails_in is None else np.ascontiguousarray(tails_in, dtype=np.int16)
        _check(self.L.dcs_decode_batch(self.h, _ptr(blob_a), blob_a.size, _ptr(srcs), srcs.size, _ptr(jobs), n,
                                       _ptr(tin), 0 if tin is None else tin.shape[0], _ptr(pcm), _ptr(err),
                                       _ptr(tails)), self.h)
        return (pcm, err, tails) if want_tails else (pcm, err)

    def call_floor(self, n_frames, iters=200):
        """dcs_ctx_call_floor -> (launch + wait, launch + copy of n_frames x 516 B + wait) in microseconds"""
        a, b = ctypes.c_float(), ctypes.c_float()
        _check(self.L.dcs_ctx_call_floor(self.h, n_frames, iters, ctypes.byref(a), ctypes.byref(b)), self.h)
        return a.value, b.value

    def decode_batch_live(self, blob, srcs, jobs, tails_in=None, blob_id=0):
        """dcs_decode_batch_live: the context's persistent small-batch decoder -> (pcm, err, tails), copies of what lies in the
        context's pinned memory until its next decode call"""
        blob_a = np.frombuffer(bytes(blob), dtype=np.uint8)
        srcs = np.ascontiguousarray(srcs, dtype=SRC_DTYPE)
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        n = jobs.size
        tin = None if tails_in is None else np.ascontiguousarray(tails_in, dtype=np.int16)
        pcm, err, tails = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        _check(self.L.dcs_decode_batch_live(self.h, _ptr(blob_a), ctypes.c_size_t(blob_a.size), ctypes.c_uint64(blob_id), _ptr(srcs), srcs.size,
                                            _ptr(jobs), n, _ptr(tin), 0 if tin is None else tin.shape[0],
                                            ctypes.byref(pcm), ctypes.byref(err), ctypes.byref(tails)), self.h)
        def view(p, dtype, shape):
            nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
            return np.frombuffer(ctypes.string_at(p.value, nbytes), dtype=dtype).reshape(shape).copy()
        return view(pcm, np.int16, (n, FRAME_SAMPLES)), view(err, np.uint32, (n,)), view(tails, np.int16, (n, 16))

    def decode_streams(self, streams, extra_frames=0):
        """streams: iterable of (os, bytes, volume, level) -> (pcm [frames,240], err, first_job)"""
        b = build_stream_batch(streams, extra_frames)
        pcm, err = self.decode_batch(b["blob"], b["srcs"], b["jobs"])
        return pcm, err, b["first_job"]

    def batch(self, blob, srcs, jobs, tails_in=None):
        return Batch(self, blob, srcs, jobs, tails_in)

    def extract_streams(self, romset, volume=255, extra_frames=2):
        """the whole `--extract-streams` pipeline: ROM set -> plan -> one launch -> PCM per stream.
        -> (plan items, pcm [frames, 240], first frame of each stream)"""
        items = romset.extract_plan()
        refs = romset.stream_refs(items, volume)
        total = ctypes.c_uint64()
        st = self.L.dcs_count_stream_frames(refs, len(items), extra_frames, ctypes.byref(total))
        if st != 0:
            raise DcsError(st)
        pcm = np.zeros((total.value, FRAME_SAMPLES), dtype=np.int16)
        first = np.zeros(len(items) + 1, dtype=np.uint32)
        _check(self.L.dcs_decode_stream_sequence(self.h, refs, len(items), extra_frames, _ptr(pcm), total.value,
                                                 _ptr(first), None), self.h)
        return items, pcm, first

    def decode_stream_sequence(self, os_, volume, streams, levels, extra_frames=2):
        """dcs_decode_stream_sequence: the --extract-streams loop on one decoder object.
        streams: list of bytes; -> (pcm [frames, 240], err, first frame of each stream)"""
        keep = [np.frombuffer(bytes(s), dtype=np.uint8) for s in streams]
        refs = (StreamRef * len(streams))()
        total = 0
        for k, s in enumerate(keep):
            refs[k].data = s.ctypes.data
            refs[k].len = s.size
            refs[k].os = os_
            refs[k].volume = volume
            refs[k].level = levels[k]
            refs[k].channelVolume = 0xFF
            total += ((int(s[0]) << 8) | int(s[1])) + extra_frames
        pcm = np.zeros((total, FRAME_SAMPLES), dtype=np.int16)
        err = np.zeros(total, dtype=np.uint32)
        first = np.zeros(len(streams) + 1, dtype=np.uint32)
        _check(self.L.dcs_decode_stream_sequence(self.h, refs, len(streams), extra_frames, _ptr(pcm), total,
                                                 _ptr(first), _ptr(err)), self.h)
        return pcm, err, first

    def extract_tracks(self, romset, plan):
        """dcs_extract_tracks: the --extract-tracks loop on one decoder, every tick planned ahead on the host sequencer,
        one launch.  plan: [(track, frames)] (RomSet.extract_tracks_plan); -> (pcm [frames, 240], first frame of each track)"""
        items = np.array(plan, dtype=np.uint32).reshape(-1, 2)
        total = int(items[:, 1].sum()) if len(plan) else 0
        pcm = np.zeros((max(total, 1), FRAME_SAMPLES), dtype=np.int16)
        first = np.zeros(len(plan) + 1, dtype=np.uint32)
        _check(self.L.dcs_extract_tracks(self.h, romset.h, _ptr(items), len(plan), _ptr(pcm), total, _ptr(first), None), self.h)
        return pcm[:total], first

    def clock_mhz(self):
        """dcs_ctx_clock_mhz: the shader clock under an integer load on every SIMD (probe kernel)"""
        mhz = ctypes.c_float()
        _check(self.L.dcs_ctx_clock_mhz(self.h, ctypes.byref(mhz)), self.h)
        return mhz.value

    def set_test_hooks(self, chunk_order_seed=0, no_xcd_ranges=False):
        """test hooks: host-planned batches get their chunks in a seeded random order; no batch is launched in XCD ranges"""
        _check(self.L.dcs_ctx_set_test_hooks(self.h, int(chunk_order_seed), int(bool(no_xcd_ranges))), self.h)

    def device_path(self, streams, extra_frames=0):
        return DevicePath(self, streams, extra_frames)

    def pipeline(self, depth=3, index_on_device=False, pack_on_device=False, plan_on_device=False):
        return Pipeline(self, depth, index_on_device, pack_on_device, plan_on_device)

    def pack_chunks_device(self, blob, srcs, jobs, fpw):
        """dcs_pack_chunks_device -> uint8 array [nChunks, packageBytes], assembled by the device packer"""
        blob_a = np.frombuffer(bytes(blob), dtype=np.uint8)
        srcs = np.ascontiguousarray(srcs, dtype=SRC_DTYPE)
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        n, pb = ctypes.c_uint32(0), ctypes.c_uint32(0)
        _check(self.L.dcs_pack_chunks_device(self.h, _ptr(jobs), jobs.size, _ptr(srcs), srcs.size, _ptr(blob_a), blob_a.size, fpw,
                                             None, 0, ctypes.byref(n), ctypes.byref(pb)), self.h)
        out = np.zeros((n.value, pb.value), dtype=np.uint8)
        _check(self.L.dcs_pack_chunks_device(self.h, _ptr(jobs), jobs.size, _ptr(srcs), srcs.size, _ptr(blob_a), blob_a.size, fpw,
                                             _ptr(out), out.size, ctypes.byref(n), ctypes.byref(pb)), self.h)
        return out

    def index_streams_gpu(self, streams):
        """dcs_index_streams_gpu: the index pass on the GPU, one wavefront per stream.  Same result as
        index_streams / index_stream."""
        streams = list(streams)
        if not streams:
            return []
        blob, locs = pack_streams(streams)
        counts = _frame_counts(streams)
        cap = int(counts.sum())
        out = np.zeros(max(cap, 1), dtype=INDEX_DTYPE)
        infos = np.zeros(len(streams), dtype=INFO_DTYPE)
        b = np.frombuffer(blob, dtype=np.uint8)
        _check(self.L.dcs_index_streams_gpu(self.h, _ptr(b), b.size, _ptr(locs), len(streams), _ptr(out), cap,
                                            _ptr(infos)), self.h)
        return _split_records(out, infos, locs["firstRecord"])

    def index_gpu_time(self, iters=10):
        ms = ctypes.c_float()
        _check(self.L.dcs_index_streams_gpu_time(self.h, iters, ctypes.byref(ms)), self.h)
        return ms.value


class Batch:
    """DcsBatch: a job list resident in HBM"""

    def __init__(self, ctx, blob, srcs, jobs, tails_in=None):
        self.ctx = ctx
        self.L = ctx.L
        blob_a = np.frombuffer(bytes(blob), dtype=np.uint8)
        srcs = np.ascontiguousarray(srcs, dtype=SRC_DTYPE)
        jobs = np.ascontiguousarray(jobs, dtype=JOB_DTYPE)
        tin = None if tails_in is None else np.ascontiguousarray(tails_in, dtype=np.int16)
        h = ctypes.c_void_p()
        _check(self.L.dcs_batch_create(ctx.h, _ptr(blob_a), blob_a.size, _ptr(srcs), srcs.size, _ptr(jobs), jobs.size,
                                       _ptr(tin), 0 if tin is None else tin.shape[0], ctypes.byref(h)), ctx.h)
        self.h = h
        self.n_jobs = jobs.size
        ctx._batches.add(self)

    def run(self, stream=None):
        _check(self.L.dcs_batch_run(self.h, ctypes.c_void_p(stream) if stream else None), self.ctx.h)

    def run_many(self, count, stream=None):
        _check(self.L.dcs_batch_run_many(self.h, ctypes.c_void_p(stream) if stream else None, int(count)), self.ctx.h)

    def time(self, iters, stream=None):
        ms = ctypes.c_float(0)
        _check(self.L.dcs_batch_time(self.h, ctypes.c_void_p(stream) if stream else None, iters, ctypes.byref(ms)),
               self.ctx.h)
        return ms.value

    @staticmethod
    def time_rotating(batches, iters, stream=None):
        """average kernel time of `iters` launches dealt round-robin to `batches` (dcs_batch_time_rotating)"""
        b0 = batches[0]
        hs = (ctypes.c_void_p * len(batches))(*[b.h for b in batches])
        ms = ctypes.c_float(0)
        _check(b0.L.dcs_batch_time_rotating(hs, len(batches), ctypes.c_void_p(stream) if stream else None, iters, ctypes.byref(ms)), b0.ctx.h)
        return ms.value

    def sync(self):
        _check(self.L.dcs_batch_sync(self.h), self.ctx.h)

    def download(self, want_tails=False):
        pcm = np.zeros((self.n_jobs, FRAME_SAMPLES), dtype=np.int16)
        err = np.zeros(self.n_jobs, dtype=np.uint32)
        tails = np.zeros((self.n_jobs, 16), dtype=np.int16) if want_tails else None
        _check(self.L.dcs_batch_download(self.h, _ptr(pcm), _ptr(err), _ptr(tails)), self.ctx.h)
        return (pcm, err, tails) if want_tails else (pcm, err)

    def download_view(self):
        """PCM and error words as numpy views of the batch's pinned host memory (no copy into Python memory);
        valid until the batch is run again or closed"""
        p, e = ctypes.c_void_p(), ctypes.c_void_p()
        _check(self.L.dcs_batch_download_view(self.h, ctypes.byref(p), ctypes.byref(e)), self.ctx.h)
        pcm = _view(p, ctypes.c_int16, np.int16, (self.n_jobs, FRAME_SAMPLES))
        err = _view(e, ctypes.c_uint32, np.uint32, (self.n_jobs,))
        return pcm, err

    @property
    def algorithmic_bytes(self):
        return int(self.L.dcs_batch_algorithmic_bytes(self.h))

    @property
    def abi_bytes(self):
        return int(self.L.dcs_batch_abi_bytes(self.h))

    @property
    def num_chunks(self):
        return int(self.L.dcs_batch_num_chunks(self.h))

    @property
    def package_bytes(self):
        """bytes of one chunk package of this batch (the kernel reads num_chunks of them)"""
        return int(self.L.dcs_batch_package_bytes(self.h))

    @property
    def frames_per_wave(self):
        return int(self.L.dcs_batch_frames_per_wave(self.h))

    def close(self):
        if self.h:
            if self.ctx.h:                      # (a closed context has already destroyed its batches)
                self.L.dcs_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipeline:
    """DcsPipeline: lists of whole streams in, PCM out in submission order, `depth` lists in flight"""

    def __init__(self, ctx, depth=3, index_on_device=False, pack_on_device=False, plan_on_device=False):
        self.ctx = ctx
        self.L = ctx.L
        h = ctypes.c_void_p()
        flags = (1 if index_on_device else 0) | (2 if pack_on_device else 0) | (4 if plan_on_device else 0)
        _check(self.L.dcs_pipeline_create(ctx.h, depth, flags, ctypes.byref(h)), ctx.h)
        self.h = h
        self._keep = []                         # (refs, byte buffers) of submitted lists, oldest first
        ctx._batches.add(self)                  # closed before the context, like a batch

    def submit_refs(self, refs, n, extra_frames=0, keep=None):
        """submit a prepared StreamRef array (see make_refs); the caller keeps it alive until collected"""
        self._keep.append((refs, keep))
        _check(self.L.dcs_pipeline_submit(self.h, refs, n, extra_frames), self.ctx.h)

    def submit(self, streams, extra_frames=0):
        streams = list(streams)
        refs, keep = _stream_refs(streams)
        self.submit_refs(refs, len(streams), extra_frames, keep)

    def collect(self):
        """-> (pcm [frames, 240], err, first frame of each stream, hostMs, deviceMs): views of pinned memory, valid
        until the next collect.  self.last_path: the stages of that list that ran on the device (bit 0 index walk, bit 1
        packer, bit 2 planner)"""
        r = PipelineResult()
        st = self.L.dcs_pipeline_collect(self.h, ctypes.byref(r))
        self.last_path = int(r.path)
        if self._keep:
            self._keep.pop(0)
        _check(st, self.ctx.h)
        pcm = _view(r.pcm, ctypes.c_int16, np.int16, (r.nFrames, FRAME_SAMPLES))
        err = _view(r.err, ctypes.c_uint32, np.uint32, (r.nFrames,))
        first = _view(r.frameOffsets, ctypes.c_uint32, np.uint32, (r.nStreams + 1,))
        return pcm, err, first, r.hostMs, r.deviceMs

    def close(self):
        if self.h:
            if self.ctx.h:
                self.L.dcs_pipeline_destroy(self.h)
            self.h = None
            self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DevicePath:
    """DcsDevicePath: a list of whole streams resident in HBM; index walk, planner, packer and decode kernels run back to
    back on the device with nothing crossing PCIe (include/dcs_hip.h dcs_device_path_*)"""

    def __init__(self, ctx, streams, extra_frames=0):
        self.ctx = ctx
        self.L = ctx.L
        streams = list(streams)
        refs, keep = _stream_refs(streams)
        h = ctypes.c_void_p()
        _check(self.L.dcs_device_path_create(ctx.h, refs, len(streams), extra_frames, ctypes.byref(h)), ctx.h)
        self.h = h
        self.n_streams = len(streams)
        self.n_frames = int(sum(((int(k[0]) << 8) | int(k[1])) + extra_frames for k in keep))
        ctx._batches.add(self)                  # closed before the context, like a batch

    def run(self, iters=10):
        """-> dict of the DcsDevicePathTimes fields: ms per pass and per kernel"""
        t = DevicePathTimes()
        _check(self.L.dcs_device_path_run(self.h, int(iters), ctypes.byref(t)), self.ctx.h)
        return {f[0]: getattr(t, f[0]) for f in DevicePathTimes._fields_}

    def run_many(self, iters):
        """`iters` passes back to back and a wait; nothing timed"""
        _check(self.L.dcs_device_path_run_many(self.h, int(iters)), self.ctx.h)

    def download(self):
        pcm = np.zeros((self.n_frames, FRAME_SAMPLES), dtype=np.int16)
        err = np.zeros(self.n_frames, dtype=np.uint32)
        first = np.zeros(self.n_streams + 1, dtype=np.uint32)
        _check(self.L.dcs_device_path_download(self.h, _ptr(pcm), _ptr(err), _ptr(first)), self.ctx.h)
        return pcm, err, first

    def close(self):
        if self.h:
            if self.ctx.h:
                self.L.dcs_device_path_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Node:
    """DcsNode: several GPUs of one node behind one object -- persistent contexts, one pipeline each, lists dealt to the
    least-loaded device, results in submission order (include/dcs_hip.h dcs_node_*)"""

    def __init__(self, device_ids, depth=8, index_on_device=True, pack_on_device=True, plan_on_device=True):
        self.L = load_library()
        devs = np.ascontiguousarray(device_ids, dtype=np.int32)
        flags = (1 if index_on_device else 0) | (2 if pack_on_device else 0) | (4 if plan_on_device else 0)
        h = ctypes.c_void_p()
        st = self.L.dcs_node_create(_ptr(devs), devs.size, depth, flags, ctypes.byref(h))
        if st != 0:
            raise DcsError(st, "dcs_node_create")
        self.h = h
        self.n_devices = int(devs.size)
        self._keep = []

    def _check(self, st):
        if st != 0:
            raise DcsError(st, self.L.dcs_node_last_error(self.h).decode(errors="replace"))

    def submit_refs(self, refs, n, extra_frames=0, keep=None):
        self._keep.append((refs, keep))
        self._check(self.L.dcs_node_submit(self.h, refs, n, extra_frames))

    def submit(self, streams, extra_frames=0):
        streams = list(streams)
        refs, keep = _stream_refs(streams)
        self.submit_refs(refs, len(streams), extra_frames, keep)

    def collect(self):
        """-> (pcm, err, first frame of each stream, hostMs, deviceMs, index of the device that decoded the list)"""
        r = PipelineResult()
        dev = ctypes.c_int(-1)
        st = self.L.dcs_node_collect(self.h, ctypes.byref(r), ctypes.byref(dev))
        if self._keep:
            self._keep.pop(0)
        self._check(st)
        self.last_path = int(r.path)
        pcm = _view(r.pcm, ctypes.c_int16, np.int16, (r.nFrames, FRAME_SAMPLES))
        err = _view(r.err, ctypes.c_uint32, np.uint32, (r.nFrames,))
        first = _view(r.frameOffsets, ctypes.c_uint32, np.uint32, (r.nStreams + 1,))
        return pcm, err, first, r.hostMs, r.deviceMs, dev.value

    def device_info(self, index):
        """-> (HIP device id, NUMA node or -1, lists decoded so far)"""
        d, nn, done = ctypes.c_int(0), ctypes.c_int(0), ctypes.c_uint64(0)
        self._check(self.L.dcs_node_device_info(self.h, index, ctypes.byref(d), ctypes.byref(nn), ctypes.byref(done)))
        return d.value, nn.value, done.value

    def close(self):
        if self.h:
            self.L.dcs_node_destroy(self.h)
            self.h = None
            self._keep = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def device_numa_node(device):
    """NUMA node of a HIP device (dcs_device_numa_node: from its PCI address), or -1"""
    return int(load_library().dcs_device_numa_node(int(device)))


def bind_process_to_device_numa(device):
    """bind the calling process (the threads it starts later inherit the mask, and the memory they pin or first touch is taken
    from that node under the default local policy) to the CPUs of the GPU's NUMA node; -> the node, or None when it is unknown
    or none of its CPUs may be used.  What dcs_node does per context, for a process that owns ONE GPU (a rank of bench.py)."""
    node = device_numa_node(device)
    if node < 0:
        return None
    try:
        text = open("/sys/devices/system/node/node%d/cpulist" % node).read().strip()
    except OSError:
        return None
    cpus = set()
    for part in text.split(","):
        if part:
            a, _, b = part.partition("-")
            cpus.update(range(int(a), int(b or a) + 1))
    allowed = cpus & os.sched_getaffinity(0)
    if not allowed:
        return None
    os.sched_setaffinity(0, allowed)
    return node


def node_cache_release():
    """destroy the contexts dcs_decode_streams_sharded keeps per device list"""
    load_library().dcs_node_cache_release()


def make_refs(streams):
    """StreamRef array for (os, bytes, volume, level) streams -> (refs, keep-alive buffers)"""
    return _stream_refs(list(streams))


class RomSet:
    """DcsRomSet: sound ROM images -> catalog, track programs, streams (include/dcs_hip.h, csrc/dcs_rom.cpp)"""

    def __init__(self, images=None, zip_path=None, zip_bytes=None, zip_name="roms.zip", explicit_u2=None):
        self.L = load_library()
        self.h = ctypes.c_void_p(self.L.dcs_romset_create())
        st = 0
        if images:
            for chip, data in images.items():
                st = st or self.L.dcs_romset_add_rom(self.h, chip, bytes(data), len(data))
        elif zip_path is not None:
            st = self.L.dcs_romset_load_zip(self.h, str(zip_path).encode(), explicit_u2.encode() if explicit_u2 else None)
        elif zip_bytes is not None:
            st = self.L.dcs_romset_load_zip_memory(self.h, zip_bytes, len(zip_bytes), zip_name.encode(),
                                                   explicit_u2.encode() if explicit_u2 else None)
        if st != 0:
            msg = self.L.dcs_romset_last_error(self.h).decode()
            self.close()
            raise DcsError(st, msg)

    def close(self):
        if self.h:
            self.L.dcs_romset_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self):
        c = RomCheck()
        _rs_check(self.L.dcs_romset_check(self.h, ctypes.byref(c)), self)
        return c

    def set_version(self, hw, os_):
        _rs_check(self.L.dcs_romset_set_version(self.h, hw, os_), self)

    @property
    def num_tracks(self):
        return self.L.dcs_romset_num_tracks(self.h)

    def track_info(self, track):
        ti = TrackInfo()
        return ti if self.L.dcs_romset_track_info(self.h, track, ctypes.byref(ti)) == 0 else None

    def decompile(self, track):
        n = ctypes.c_uint32()
        _rs_check(self.L.dcs_romset_decompile(self.h, track, None, 0, ctypes.byref(n)), self)
        ops = np.zeros(max(n.value, 1), dtype=TRACKOP_DTYPE)
        _rs_check(self.L.dcs_romset_decompile(self.h, track, _ptr(ops), n.value, ctypes.byref(n)), self)
        return ops[:n.value]

    def list_streams(self):
        n = ctypes.c_uint32()
        _rs_check(self.L.dcs_romset_list_streams(self.h, None, 0, ctypes.byref(n)), self)
        a = np.zeros(max(n.value, 1), dtype=np.uint32)
        _rs_check(self.L.dcs_romset_list_streams(self.h, _ptr(a), n.value, ctypes.byref(n)), self)
        return a[:n.value]

    def pointer(self, linear):
        """-> (chip number 2..9, offset in that chip's image, bytes available from there, address)"""
        p, avail, chip = ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_int()
        _rs_check(self.L.dcs_romset_pointer(self.h, linear, ctypes.byref(p), ctypes.byref(avail), ctypes.byref(chip)), self)
        return chip.value, p.value, avail.value

    def stream_bytes(self, linear):
        _, p, avail = self.pointer(linear)
        return ctypes.string_at(p, avail)

    def extract_plan(self):
        n = ctypes.c_uint32()
        _rs_check(self.L.dcs_romset_extract_plan(self.h, None, 0, ctypes.byref(n)), self)
        a = np.zeros(max(n.value, 1), dtype=EXTRACT_DTYPE)
        _rs_check(self.L.dcs_romset_extract_plan(self.h, _ptr(a), n.value, ctypes.byref(n)), self)
        return a[:n.value]

    def extract_tracks_plan(self):
        """dcs_romset_extract_tracks_plan: [(track, frames of its WAV file)] of the --extract-tracks loop"""
        n = ctypes.c_uint32()
        _rs_check(self.L.dcs_romset_extract_tracks_plan(self.h, None, 0, ctypes.byref(n)), self)
        a = np.zeros((max(n.value, 1), 2), dtype=np.uint32)
        _rs_check(self.L.dcs_romset_extract_tracks_plan(self.h, _ptr(a), n.value, ctypes.byref(n)), self)
        return [(int(t), int(f)) for t, f in a[:n.value]]

    def add_rom(self, chip, data):
        _rs_check(self.L.dcs_romset_add_rom(self.h, chip, bytes(data), len(data)), self)

    def stream_refs(self, items, volume):
        refs = (StreamRef * max(len(items), 1))()
        items = np.ascontiguousarray(items, dtype=EXTRACT_DTYPE)
        _rs_check(self.L.dcs_romset_stream_refs(self.h, _ptr(items), len(items), volume, refs), self)
        return refs

    def dump(self, force_hw=-1, force_os=-1):
        """everything derived from the images as text, one item per line (the tests compare it with the same dump
        made from the reference's answers)"""
        c = self.check()
        out = ["check status=%d hw=%d os=%d nominal=%04x catalog=%x ntracks=%u sig=%s" % (
            c.status, c.hw, c.os, c.nominalVersion, c.catalogOffset, c.nTracks, c.signature.decode())]
        if force_hw >= 0 or force_os >= 0:
            self.set_version(force_hw if force_hw >= 0 else c.hw, force_os if force_os >= 0 else c.os)
        for t in range(self.num_tracks):
            ti = self.track_info(t)
            if ti is None:
                continue
            out.append("track %u addr=%06x ch=%d type=%d defer=%04x time=%u loop=%d" % (
                t, ti.address, ti.channel, ti.type, ti.deferCode & 0xFFFF, ti.time, ti.looping))
            if ti.type != 1:
                continue
            for op in self.decompile(t):
                out.append(" op off=%d nest=%d parent=%d delay=%04x opc=%02x n=%d bytes=%s" % (
                    op["offset"], op["nestingLevel"], op["loopParent"], op["delayCount"], op["opcode"], op["nOperandBytes"],
                    bytes(op["operandBytes"][:min(int(op["nOperandBytes"]), 8)]).hex()))
        base = {}
        for a in self.list_streams():
            chip, p, avail = self.pointer(int(a))
            if chip not in base:
                base[chip] = self.pointer((chip - 2) << (21 if self.check_hw() == HW_DCS95 else 20))[1]
            out.append("stream %06x chip=%d off=%x" % (a, chip, p - base[chip]))
        for it in self.extract_plan():
            out.append("extract track=%u num=%d addr=%06x level=%d" % (it["track"], it["streamNum"], it["address"], it["level"]))
        return "\n".join(out) + "\n"

    def check_hw(self):
        # the version in effect (after a possible override) shows in how a pointer is split
        chip, _, _ = self.pointer(1 << 20)
        return HW_DCS93 if chip == 3 else HW_DCS95


def _rs_check(st, rs):
    if st != 0:
        raise DcsError(st, rs.L.dcs_romset_last_error(rs.h).decode())


class Sequencer:
    """DcsSequencer: the track-program VM in front of the frame decode (csrc/dcs_sequencer.cpp)"""

    def __init__(self, romset, volume=255):
        self.L = load_library()
        self.rs = romset                        # keep the ROM set alive
        self.h = ctypes.c_void_p(self.L.dcs_seq_create(romset.h))
        if not self.h:
            raise DcsError(ERR_INVALID_ARG, "dcs_seq_create: ROM set without U2 or without versions")
        self.L.dcs_seq_set_master_volume(self.h, volume)

    def close(self):
        if self.h:
            self.L.dcs_seq_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def event(self, kind, value):
        """kind 0 data-port byte, 1 track command, 2 master volume, 3 ClearTracks"""
        if kind == 0: self.L.dcs_seq_write_data_port(self.h, value)
        elif kind == 1: self.L.dcs_seq_add_track_command(self.h, value)
        elif kind == 2: self.L.dcs_seq_set_master_volume(self.h, value)
        elif kind == 3: self.L.dcs_seq_clear_tracks(self.h)

    def set_rewindable(self, on=True):
        st = self.L.dcs_seq_set_rewindable(self.h, 1 if on else 0)
        if st != 0:
            raise DcsError(st)

    def rewind(self, keep_ticks):
        st = self.L.dcs_seq_rewind(self.h, keep_ticks)
        if st != 0:
            raise DcsError(st)

    def run_script(self, n_ticks, events):
        """plan n_ticks ticks, applying the (tick, kind, value) events before their tick"""
        e = 0
        events = sorted(events, key=lambda x: x[0])
        for t in range(n_ticks):
            while e < len(events) and events[e][0] <= t:
                self.event(events[e][1], events[e][2])
                e += 1
            st = self.L.dcs_seq_plan(self.h, 1)
            if st != 0:
                raise DcsError(st, self.L.dcs_seq_last_error(self.h).decode())

    def plan(self, n_ticks):
        st = self.L.dcs_seq_plan(self.h, n_ticks)
        if st != 0:
            raise DcsError(st, self.L.dcs_seq_last_error(self.h).decode())

    def plan_ahead(self, max_ticks, idle_ticks=2):
        """dcs_seq_plan_ahead -> ticks planned"""
        n = ctypes.c_uint32()
        st = self.L.dcs_seq_plan_ahead(self.h, max_ticks, idle_ticks, ctypes.byref(n))
        if st != 0:
            raise DcsError(st, self.L.dcs_seq_last_error(self.h).decode())
        return n.value

    def stream_playing_at(self, ticks, channel):
        return bool(self.L.dcs_seq_stream_playing_at(self.h, ticks, channel))

    def tracks_active_at(self, ticks):
        return bool(self.L.dcs_seq_tracks_active_at(self.h, ticks))

    def stream_playing(self, channel):
        return bool(self.L.dcs_seq_stream_playing(self.h, channel))

    def decode_view(self, ctx):
        """dcs_seq_decode_view: (pcm, err) copied out of the context's pinned memory"""
        pcm, err, n = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_uint32()
        _check(self.L.dcs_seq_decode_view(ctx.h, self.h, ctypes.byref(pcm), ctypes.byref(n), ctypes.byref(err)), ctx.h)
        if n.value == 0:
            return np.zeros((0, FRAME_SAMPLES), dtype=np.int16), np.zeros(0, dtype=np.uint32)
        p = np.frombuffer(ctypes.string_at(pcm.value, n.value * FRAME_SAMPLES * 2), dtype=np.int16).reshape(n.value, FRAME_SAMPLES).copy()
        e = np.frombuffer(ctypes.string_at(err.value, n.value * 4), dtype=np.uint32).copy()
        return p, e

    @property
    def pending_ticks(self):
        return self.L.dcs_seq_pending_ticks(self.h)

    @property
    def fatal(self):
        return bool(self.L.dcs_seq_is_fatal(self.h))

    def host_bytes(self):
        n = self.L.dcs_seq_host_bytes(self.h, None, 0)
        a = np.zeros((max(n, 1), 2), dtype=np.uint32)
        self.L.dcs_seq_host_bytes(self.h, _ptr(a), n)
        return [(int(t), int(b)) for t, b in a[:n]]

    def decode(self, ctx):
        n = self.pending_ticks
        pcm = np.zeros((n, FRAME_SAMPLES), dtype=np.int16)
        err = np.zeros(max(n, 1), dtype=np.uint32)
        _check(self.L.dcs_seq_decode(ctx.h, self.h, _ptr(pcm), n, _ptr(err)), ctx.h)
        return pcm, err[:n]
