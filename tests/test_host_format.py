"""Host writers: the string formatters used by saveAll() equal the iostream formatters byte for byte."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fast_formatters_match_iostream(tmp_path):
    host = os.path.join(ROOT, "portcullis_amd", "host")
    csrc = os.path.join(ROOT, "portcullis_amd", "csrc")
    exe = str(tmp_path / "fmt")
    subprocess.check_call(["g++", "-O1", "-std=c++17", f"-I{host}/include", f"-I{ROOT}/include", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "format_equivalence.cc"), f"-L{host}", "-lportcullis_host",
                           f"-L{csrc}", "-lportcullis_amd", f"-Wl,-rpath,{host}", f"-Wl,-rpath,{csrc}"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout + out.stderr


def test_formatters_under_sanitizers(tmp_path):
    """The same driver with junction.cc / junction_system.cc / genome_mapper.cc / bam_reader.cc / bam_writer.cc compiled under
    AddressSanitizer + UndefinedBehaviorSanitizer (the device library is only linked: nothing here calls it)."""
    host = os.path.join(ROOT, "portcullis_amd", "host")
    csrc = os.path.join(ROOT, "portcullis_amd", "csrc")
    exe = str(tmp_path / "fmt_asan")
    src = [os.path.join(host, "src", f) for f in ("junction.cc", "junction_system.cc", "genome_mapper.cc", "bam_reader.cc", "fast_inflate.cc", "bam_writer.cc")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", f"-I{host}/include", f"-I{ROOT}/include", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "format_equivalence.cc")] + src +
                          [f"-L{csrc}", "-lportcullis_amd", f"-Wl,-rpath,{csrc}", "-lz", "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout + out.stderr[-2000:]


def test_absorb_equals_append(tmp_path):
    """JunctionSystem::absorb (findJunctions' merge: the per-target maps' nodes move over) against append on random per-target systems,
    with and without introns in common; calcJunctionStats on both merged lists gives the same table."""
    host = os.path.join(ROOT, "portcullis_amd", "host")
    csrc = os.path.join(ROOT, "portcullis_amd", "csrc")
    exe = str(tmp_path / "absorb")
    src = [os.path.join(host, "src", f) for f in ("junction.cc", "junction_system.cc", "genome_mapper.cc", "bam_reader.cc", "fast_inflate.cc", "bam_writer.cc")]
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-fno-omit-frame-pointer", f"-I{host}/include", f"-I{ROOT}/include", "-o", exe,
                           os.path.join(ROOT, "tests", "cpp", "absorb_equivalence.cc")] + src +
                          [f"-L{csrc}", "-lportcullis_amd", f"-Wl,-rpath,{csrc}", "-lz", "-lpthread"])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "mismatches: 0" in out.stdout, out.stdout + out.stderr[-2000:]
