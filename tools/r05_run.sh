SKIP_CHECKS=1 bash tools/r05_ab.sh glow; bash tools/r05_trace.sh noglow plainflag glow
