/* TEST-ONLY stand-in for enet/enet.h: the types, constants and prototypes /root/reference/src/main.c names (see raylib.h in
 * the parent directory).  Written from ENet's public API as main.c uses it; not ENet. */
#pragma once
#include <stddef.h>
#include <stdint.h>
typedef uint8_t enet_uint8;
typedef uint16_t enet_uint16;
typedef uint32_t enet_uint32;
typedef struct _ENetAddress { enet_uint32 host; enet_uint16 port; } ENetAddress;
typedef struct _ENetPacket { size_t referenceCount; enet_uint32 flags; enet_uint8 *data; size_t dataLength; } ENetPacket;
typedef struct _ENetPeer { ENetAddress address; void *data; enet_uint32 connectID; enet_uint16 incomingPeerID; } ENetPeer;
typedef struct _ENetHost { ENetAddress address; ENetPeer *peers; size_t peerCount; size_t connectedPeers; } ENetHost;
typedef enum _ENetEventType { ENET_EVENT_TYPE_NONE = 0, ENET_EVENT_TYPE_CONNECT = 1, ENET_EVENT_TYPE_DISCONNECT = 2, ENET_EVENT_TYPE_RECEIVE = 3 } ENetEventType;
typedef struct _ENetEvent { ENetEventType type; ENetPeer *peer; enet_uint8 channelID; enet_uint32 data; ENetPacket *packet; } ENetEvent;
enum { ENET_PACKET_FLAG_RELIABLE = 1, ENET_PACKET_FLAG_UNSEQUENCED = 2 };
#define ENET_HOST_ANY 0
int enet_initialize(void);
void enet_deinitialize(void);
int enet_address_set_host(ENetAddress *address, const char *hostName);
ENetHost *enet_host_create(const ENetAddress *address, size_t peerCount, size_t channelLimit, enet_uint32 incomingBandwidth, enet_uint32 outgoingBandwidth);
void enet_host_destroy(ENetHost *host);
ENetPeer *enet_host_connect(ENetHost *host, const ENetAddress *address, size_t channelCount, enet_uint32 data);
int enet_host_service(ENetHost *host, ENetEvent *event, enet_uint32 timeout);
void enet_host_flush(ENetHost *host);
void enet_host_broadcast(ENetHost *host, enet_uint8 channelID, ENetPacket *packet);
ENetPacket *enet_packet_create(const void *data, size_t dataLength, enet_uint32 flags);
void enet_packet_destroy(ENetPacket *packet);
int enet_peer_send(ENetPeer *peer, enet_uint8 channelID, ENetPacket *packet);
void enet_peer_disconnect(ENetPeer *peer, enet_uint32 data);
void enet_peer_reset(ENetPeer *peer);
