P="timeout -k 10 150 python tools/stackw_probe.py"
QPN_STACK_WAVE_FWD=2 $P fwd_h_noscratch
QPN_STACK_WAVE_FWD=2 QPN_STACK_WGS=448 $P fwd_h_448
QPN_STACK_WAVE_FWD=2 QPN_STACK_WGS=384 $P fwd_h_384
QPN_STACK_WAVE_FWD=2 QPN_STACK_WGS=640 $P fwd_h_640
$P fwd_old
