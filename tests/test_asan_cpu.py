"""Host C++ under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build; SURVEY.md §5 assigns sanitizer coverage to the build - the
reference has none).  tests/asan/build_asan.sh instruments errors.cpp + the six host sources and links them with the product's HIP
objects; the host-side CPU tests then run against that library in a child interpreter with libasan preloaded: golden G6 x 24 and G10
(polyline assembly), the endpoint clustering cases, G13 + six random roads (cross-tile merge), G12 (back-projection), the PNG corpus
incl. the damaged-file fuzz cases, the DEFLATE decoder on every block type and on 4,500 damaged / truncated streams, the LAS header parser, both JSON writers (4.7 MB of float bit patterns) and the skeleton cases."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_TESTS = ('test_cpp_polyline_assembly_vs_reference_golden or test_cpp_assembly_on_reference_e2e_decode or test_cpp_endpoint_clustering_vs_oracle '
              'or test_semantic_raster_vs_oracle or test_empty_and_degenerate_tiles or test_json_output_is_byte_identical_to_reference '
              'or test_cpp_trace_lines_vs_rowref_golden or test_polyline_backproject_golden_g12 or test_polyline_backproject_errors_and_roundtrip '
              'or test_las_header_parse or test_png_reader_matches_pil or test_png_reader_hand_filtered_rows_and_errors '
              'or test_png_reader_survives_damaged_files or test_zlib_inflate_matches_zlib_on_every_block_type or test_zlib_inflate_refuses_malformed_streams '
              'or test_native_lane_json_is_json_dump_byte_for_byte '
              'or test_native_seqs_json_is_json_dump_byte_for_byte or test_merge_lines_golden_g13 or test_merge_lines_cpp_vs_oracle_random_roads '
              'or test_skeleton_cpp_vs_3d_oracle or test_skeleton_properties or test_skeleton_known_answers or test_skeleton_published_shapes or test_line8_opencv_table or test_eval_metric_line_segmentor')


def test_host_cpp_clean_under_asan_ubsan():
    if shutil.which('g++') is None:
        pytest.skip('no g++')
    asan = subprocess.run(['gcc', '-print-file-name=libasan.so'], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip('no libasan')
    b = subprocess.run(['bash', os.path.join(ROOT, 'tests', 'asan', 'build_asan.sh')], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert b.returncode == 0, (b.stdout + b.stderr)[-3000:]
    lib = os.path.join(ROOT, 'tests', 'asan', '_build', 'libhost_asan.so')
    env = dict(os.environ, LD_PRELOAD=asan, LANEMAP_HIP_LIB=lib,
               # (python itself leaks by design; torch's allocator pairs new / free across libraries)
               ASAN_OPTIONS='detect_leaks=0:alloc_dealloc_mismatch=0:detect_odr_violation=0:abort_on_error=0:exitcode=97',
               UBSAN_OPTIONS='print_stacktrace=1:halt_on_error=1:exitcode=98')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(ROOT, 'tests', 'test_boundary_cpu.py'),
                        os.path.join(ROOT, 'tests', 'test_metrics_io_cpu.py'), '-x', '-q', '-p', 'no:cacheprovider', '-k', HOST_TESTS],
                       capture_output=True, text=True, timeout=1500, cwd=ROOT, env=env)
    out = r.stdout + r.stderr
    assert 'AddressSanitizer' not in out and 'runtime error:' not in out, out[-4000:]
    assert r.returncode == 0 and ' passed' in r.stdout, out[-3000:]
    n = int(r.stdout.strip().split('\n')[-1].split(' passed')[0].split()[-1])
    assert n >= 23, f'only {n} host tests ran under the sanitizers'
