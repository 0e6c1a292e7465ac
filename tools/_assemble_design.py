"""one-off: DESIGN.md re-cut for round 4 (current-state sections first, the round-by-round history in an appendix).
Reads the round-3 DESIGN.md from git (94893c2) and the new sections from gpurun_out/design_*.md."""
import subprocess

old = subprocess.check_output(["git", "show", "94893c2:DESIGN.md"], text=True).split("\n")


def L(a, b):  # 1-based inclusive line range of the old file
    return "\n".join(old[a - 1:b])


def new(name):
    return open("gpurun_out/design_%s.md" % name).read().rstrip("\n")


parts = []
parts.append(new("head"))            # title, section 1
parts.append(L(40, 88))              # section 2 (arithmetic contract)
parts.append(new("s2_add"))          # the clamp as a documented limit
parts.append("")
parts.append(new("s3"))              # section 3 data layout
parts.append("")
parts.append(L(109, 147))            # 4, 4.1
parts.append(new("s4_2"))            # 4.2, 4.3 current state
parts.append("")
parts.append(new("s4_4"))            # 4.4 with the zero rule closed and the ring bound under a far d_max
parts.append("")
parts.append(L(382, 566))            # 4.5 .. 4.7 + hardware mapping paragraph
parts.append(new("4_8"))
parts.append("")
parts.append(new("s5"))              # oracle and parity
parts.append("")
parts.append(new("s6"))              # measurement
parts.append("")
parts.append(new("s7"))              # multi-GPU, what comes next
parts.append("")
parts.append(L(886, 902))            # 8, 9
parts.append("")
parts.append(new("appendix_head"))
parts.append("### A.1 The scoring kernels: what was tried (rounds 1 - 3; section 4.3 until round 3)\n")
parts.append(L(186, 194))
parts.append("")
parts.append(L(213, 337))
parts.append("\n### A.2 Measurement history (rounds 1 - 3; section 6 until round 3)\n")
parts.append(L(626, 787))
parts.append("\n### A.3 Multi-GPU estimates and the ranked lists of rounds 2 - 3 (section 7 until round 3)\n")
parts.append(L(814, 884))
text = "\n".join(parts) + "\n"
for old_s, new_s in (("(`NuisWorker`, `ig_hip.hip`:", "(`NuisWorker`, `csrc/ig_host_core.inc`:"),
                     ("**The step is now bound by the host**, so the host side changed with it:",
                      "**With it the step was bound by the host** (round 3; round 4 moved whole runs of pairs onto the device: §4.8), so the host side changed with it:")):
    assert text.count(old_s) == 1, old_s
    text = text.replace(old_s, new_s)
open("DESIGN.md", "w").write(text)
print("DESIGN.md:", sum(len(p.split("\n")) for p in parts), "lines")
