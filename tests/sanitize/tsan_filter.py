"""Condense a ThreadSanitizer log: warnings whose racing accesses (the '#0' frames of the access stacks) are in this
repository's code, as opposed to inside uninstrumented libraries (the HIP runtime) that merely sit under our calls."""
import re
import sys

text = open(sys.argv[1], errors="replace").read()
blocks = re.split(r"(?==================\nWARNING: ThreadSanitizer)", text)
ours = 0
for b in blocks:
    if "WARNING: ThreadSanitizer" not in b:
        continue
    # access stacks: sections that start with "Write of size" / "Read of size" / "Previous write" / "Previous read" / atomic
    heads = [m.start() for m in re.finditer(r"^\s+(Previous )?(atomic )?(write|read|Write|Read) of size", b, re.M)]
    top_frames = []
    for h in heads:
        m = re.search(r"^\s+#0 (.*)$", b[h:], re.M)
        if m:
            top_frames.append(m.group(1))
    if top_frames and all("/root/repo/" in f or "hugectr_backend_amd" in f for f in top_frames):
        ours += 1
        print("---- race with both accesses in repository code ----")
        for f in top_frames:
            print("   ", f[:200])
print(f"warnings with both accesses in repository code: {ours} (of {sum('WARNING: ThreadSanitizer' in b for b in blocks)})")
