package net.consensys.wittgenstein.core;

import java.util.List;

/**
 * The two places where core.gpu.GpuNetwork needs what the reference keeps package-private / protected: the Node counters
 * that receiveUntil and createMessageArrival bump (C/Network.java:476-477, 611-612; C/Node.java:75-79 `protected long`)
 * and the partition cuts (C/Network.java:45 `final List<Integer> partitionsInX`).
 */
public final class GpuNodeAccess {
  private GpuNodeAccess() {}

  public static void received(Node to, int bytes) {
    to.msgReceived++;
    to.bytesReceived += bytes;
  }

  public static void sent(Node from, int msgs, long bytes) {
    from.msgSent += msgs;
    from.bytesSent += bytes;
  }

  public static List<Integer> partitionsInX(Network<?> n) {
    return n.partitionsInX;
  }
}
