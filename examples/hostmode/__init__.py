"""Host-callback-mode examples: reference protocols whose Message.action() stays with the caller (wg_next_delivery,
include/wittgpu.h) while the queue, its ordering, latency sampling and rd live in the engine. They stand in for the
Java caller of that mode (class, field and method names follow P/CasperIMD.java, P/SanFerminSignature.java +
P/SanFerminHelper.java, C/P2PNetwork.java + C/messages/FloodMessage.java) and are NOT part of the product package:
tests drive them against the oracle to show the mode carries unmodified protocols."""
