P="timeout -k 10 150 python tools/stackw_probe.py"
$P base
QPN_CAUSAL_SIDE=1 $P causal_side
QPN_CAUSAL_SIDE=1 QPN_WGRAD_CHUNKS=48 $P causal_side_c48
QPN_WGRAD_CHUNKS=48 $P c48
QPN_STACK_WGS_BWD=384 $P bwd384
QPN_STACK_WGS_BWD=256 $P bwd256
